// msm_bench -- the reference's MSM benchmark CLI (examples/msm_bench.rs) over the MI355X library.
//
//   msm_bench <n> [-c|--curve pallas|vesta] [-m|--msm naive|boscoster|pippenger] [--device D]
//
// Same input recipe (one ark_std::test_rng() stream: n points by G::rand, then n scalars,
// msm_bench.rs:40-44), same timed region (only the msm call, host vectors in -> point out, :45-48), same
// output (seconds on stdout, :78).  `pippenger` is util::msm = bpgpu_msm; `naive` (the reference's default,
// :18-20) is n independent scalar multiplications on the device plus their sum; `boscoster` is a serial
// heap algorithm (src/util/msm.rs:34-81) with no data-parallel form: it runs on the host, as in the reference.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../bulletproofs_amd/host/ark_bp.hpp"

using namespace ark_bp;

template <int CURVE>
double msm_bench(size_t n, const std::string& alg, bool print_point) {
  StdRng rng = test_rng();
  std::vector<G> gs = rand_points<CURVE>(rng, n);  // square roots of the candidates on the device
  std::vector<Fr> xs;
  xs.reserve(n);
  for (size_t i = 0; i < n; i++) xs.push_back(rand_scalar<CURVE>(rng));
  (void)util::msm<CURVE>(gs.data(), n < 64 ? n : 64, xs.data(), n < 64 ? n : 64);  // context warm-up, untimed like the sampling
  auto start = std::chrono::steady_clock::now();
  G r = alg == "naive" ? util::naive_msm<CURVE>(gs, xs)
        : alg == "boscoster" ? util::bos_coster_msm<CURVE>(gs, xs)  // serial heap algorithm on the host, as in the reference
                             : util::msm<CURVE>(gs, xs);
  auto end = std::chrono::steady_clock::now();
  if (print_point) {
    uint8_t b[96];
    PointOps<CURVE>::to_bytes(r, b);
    for (int i = 0; i < 96; i++) std::fprintf(stderr, "%02x", b[i]);
    std::fprintf(stderr, "\n");
  }
  return std::chrono::duration<double>(end - start).count();
}

int main(int argc, char** argv) {
  std::string curve = "pallas", alg = "naive";
  long n = -1;
  int device = 0;
  bool print_point = false;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto value = [&](const char* name) -> std::string {
      if (i + 1 >= argc) {
        std::fprintf(stderr, "error: %s needs a value\n", name);
        std::exit(2);
      }
      return argv[++i];
    };
    if (a == "-c" || a == "--curve") curve = value("--curve");
    else if (a == "-m" || a == "--msm") alg = value("--msm");
    else if (a == "--device") device = std::atoi(value("--device").c_str());
    else if (a == "--print-point") print_point = true;  // canonical X | Y | Z of the result on stderr (tests)
    else if (!a.empty() && a[0] != '-' && n < 0) n = std::atol(a.c_str());
    else {
      std::fprintf(stderr, "usage: msm_bench <n> [-c pallas|vesta] [-m naive|boscoster|pippenger] [--device D]\n");
      return 2;
    }
  }
  for (auto& ch : curve) ch = (char)std::tolower(ch);
  for (auto& ch : alg) ch = (char)std::tolower(ch);
  if (n < 0 || (curve != "pallas" && curve != "vesta") || (alg != "naive" && alg != "boscoster" && alg != "pippenger")) {
    std::fprintf(stderr, "usage: msm_bench <n> [-c pallas|vesta] [-m naive|boscoster|pippenger] [--device D]\n");
    return 2;
  }
  try {
    check(bpgpu_init(device));
    double s = curve == "pallas" ? msm_bench<0>((size_t)n, alg, print_point) : msm_bench<1>((size_t)n, alg, print_point);
    std::printf("%g\n", s);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "panicked: %s\n", e.what());
    return 101;
  }
  bpgpu_shutdown();
  return 0;
}
