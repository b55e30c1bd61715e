// test_ipa -- the reference's IPA harness (src/lib.rs:110-135, as driven by `examples/llbp.rs -p bp` and the
// tests at src/lib.rs:253-265) for the classic Bulletproofs prover `Bp` over the MI355X library.
//
//   test_ipa <size> [<size> ...] [-r|--reps R] [-c|--curve pallas|vesta] [--device D]
//
// Per (size, repetition), from one ark_std::test_rng() stream: IpaInstance::sample_from_length, IpaRelation::check,
// 32-byte Fiat-Shamir seed, Bp::prove (timed), proof size, Bp::verify (timed) -- the three figures the
// reference logs under pf_time / pf_size / ver_time (src/lib.rs:122-132).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../bulletproofs_amd/host/ark_bp.hpp"

using namespace ark_bp;

template <int CURVE>
void test_ipa(const std::vector<size_t>& sizes, int reps) {
  StdRng rng = test_rng();
  for (size_t size : sizes) {
    for (int r = 0; r < reps; r++) {
      auto iw = sample_from_length<CURVE>(rng, size);
      IpaRelation<CURVE>::check(iw.first, iw.second);
      std::array<uint8_t, 32> fs_seed = gen_seed32(rng);
      FiatShamirRng fs = FiatShamirRng::from_seed(fs_seed.data(), 32);
      auto t0 = std::chrono::steady_clock::now();
      BpProof proof = Bp<CURVE>::prove(iw.first, iw.second, fs);
      auto t1 = std::chrono::steady_clock::now();
      std::printf("size %zu  Proof time (s): %g\n", size, std::chrono::duration<double>(t1 - t0).count());
      std::printf("size %zu  Proof size: %zu\n", size, Bp<CURVE>::proof_size(proof));
      FiatShamirRng fs2 = FiatShamirRng::from_seed(fs_seed.data(), 32);
      auto t2 = std::chrono::steady_clock::now();
      Bp<CURVE>::verify(iw.first, proof, fs2);
      auto t3 = std::chrono::steady_clock::now();
      std::printf("size %zu  Verifier time (s): %g\n", size, std::chrono::duration<double>(t3 - t2).count());
    }
  }
}

int main(int argc, char** argv) {
  std::vector<size_t> sizes;
  std::string curve = "pallas";
  int reps = 1, device = 0;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    if ((a == "-r" || a == "--reps") && i + 1 < argc) reps = std::atoi(argv[++i]);
    else if ((a == "-c" || a == "--curve") && i + 1 < argc) curve = argv[++i];
    else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
    else if (!a.empty() && a[0] != '-') sizes.push_back((size_t)std::atol(a.c_str()));
    else {
      std::fprintf(stderr, "usage: test_ipa <size>... [-r reps] [-c pallas|vesta] [--device D]\n");
      return 2;
    }
  }
  if (sizes.empty() || (curve != "pallas" && curve != "vesta")) {
    std::fprintf(stderr, "usage: test_ipa <size>... [-r reps] [-c pallas|vesta] [--device D]\n");
    return 2;
  }
  try {
    check(bpgpu_init(device));
    if (curve == "pallas") test_ipa<0>(sizes, reps); else test_ipa<1>(sizes, reps);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "panicked: %s\n", e.what());
    return 101;
  }
  bpgpu_shutdown();
  return 0;
}
