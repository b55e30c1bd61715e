"""The reference's two command-line drivers for this path, rebuilt over the library
(examples/msm_bench.cpp, examples/test_ipa.cpp): same arguments, same sampling recipe, same output.
The MSM result of the faithful recipe (examples/msm_bench.rs:40-44) is compared with the oracle's."""
import os
import subprocess

import pytest

from oracle import c_oracle as co
from oracle import pasta_model as pm
from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bulletproofs_amd", "bin")


def run(*argv):
    return subprocess.run([os.path.join(BIN, argv[0]), *map(str, argv[1:])], capture_output=True, text=True, timeout=110)


@pytest.mark.parametrize("c", [pm.PALLAS, pm.VESTA], ids=lambda c: c.name)
@pytest.mark.parametrize("alg", ["pippenger", "naive", "boscoster"])
def test_msm_bench_cli(c, alg):
    n = 300
    r = run("msm_bench", n, "-c", c.name.lower(), "-m", alg, "--print-point")
    assert r.returncode == 0, r.stderr
    assert float(r.stdout.strip()) > 0.0                       # seconds on stdout, msm_bench.rs:78
    raw = bytes.fromhex(r.stderr.strip().splitlines()[-1])      # canonical X | Y | Z
    X, Y, Z = (int.from_bytes(raw[i:i + 32], "little") for i in (0, 32, 64))
    pts, sc = co.msm_bench_inputs(c.curve_id, n)
    want = H.jac_from_abi(c, co.msm_pippenger(c.curve_id, pts, sc))[0]
    assert Z == 1 and (X, Y) == want


def test_msm_bench_cli_usage():
    assert run("msm_bench").returncode == 2
    assert run("msm_bench", 10, "-m", "heap").returncode == 2
    assert run("msm_bench", 10, "-c", "bls").returncode == 2
    r = run("msm_bench", 0, "-m", "pippenger")                          # empty MSM is the identity, no panic
    assert r.returncode == 0, r.stderr


def test_test_ipa_cli():
    """sizes of the reference's own tests (src/lib.rs:188,257); verify panics on failure (exit 101)."""
    r = run("test_ipa", 1, 2, 4, 8, 16, 17, "-r", 2)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 6 * 2 * 3
    sizes = [int(l.split()[-1]) for l in lines if "Proof size" in l]
    assert sizes == [2, 2, 4, 4, 6, 6, 8, 8, 10, 10, 12, 12]            # 2 log2(padded n) + 2, bp_iter.rs:187-189


def test_static_archive_links_and_agrees():
    """libbpgpu.a (the archive the reference's build.rs links) carries the same code object:
    msm_bench linked statically against it prints the same point as the shared-library build."""
    outs = []
    for exe in ("msm_bench", "msm_bench_static"):
        r = run(exe, 300, "-m", "pippenger", "--print-point")
        assert r.returncode == 0, r.stderr
        outs.append(r.stderr.strip().splitlines()[-1])
    assert outs[0] == outs[1] and len(outs[0]) == 192
