"""The compiler's 24-bit integer-division expansion is wrong for full 24-bit numerators ("the lost subset"; reproducer and host
model: tools/micro/urem24.hip).  CPU: no kernel of the product contains that expansion (tools/check_div24.py scans the gfx950 code of every
object), and the scanner does find it in the reproducer.  GPU: the reproducer behaves as the host model of the instruction sequence predicts."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="needs the ROCm LLVM tools")
def test_no_kernel_of_the_product_divides_24_bit_operands(tmp_path):
    import check_div24
    from metaeuk_amd import build
    build.build()
    objs, found = check_div24.scan()
    assert len(objs) >= 8
    assert not found, found
    # the scanner itself: the reproducer's kernel has the sequence
    obj = str(tmp_path / "urem24.o")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-c", os.path.join(ROOT, "tools", "micro", "urem24.hip"), "-o", obj],
                          stderr=subprocess.DEVNULL)
    _, found = check_div24.scan([obj])
    assert len(found) == 1 and "rem24_kernel" in found[0][1]


@pytest.mark.gpu
def test_the_division_reproducer_behaves_as_modelled():
    exe = os.path.join(ROOT, "tools", "micro", "_build", "urem24")
    if not os.path.exists(exe):
        pytest.skip("tools/micro/_build/urem24 not built")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    # 0: a compiler that divides correctly; 3: wrong remainders, every one of them with true remainder n - 1 in the upper range of the numerators
    assert r.returncode in (0, 3), out
    if r.returncode == 3:
        assert "n = 11: device wrong for" in out and "n = 16: device wrong for        0" in out, out


def test_host_model_of_the_division_sequence_matches_what_the_device_returned():
    """The emitted sequence -- fq = trunc(float(y) * rcp(float(n))), fr = fma(-fq, n, y), q = fq + (|fr| >= n), r = (y - q n) & 0xFFFFFF -- modelled in numpy
    float32 with a correctly rounded reciprocal: for n = 11, 44, 46, 57 it returns a wrong remainder for exactly as many of the 2^24 numerators as the
    MI355X did (profiles/r05_urem24_reproducer.txt), every one of them with true remainder n - 1 in the upper part of the range; for powers of two
    and for 10 (whose reciprocal rounds down) it is exact."""
    import numpy as np
    y = np.arange(1 << 24, dtype=np.uint32)
    fa = y.astype(np.float32)
    device_counts = {11: 476625, 44: 119156, 46: 36473, 57: 32193, 2: 0, 10: 0, 16: 0, 64: 0, 80: 0}
    for n, expected in device_counts.items():
        fb = np.float32(n)
        rc = np.float32(1.0) / fb
        fq = np.trunc(fa * rc).astype(np.float32)
        fr = fa.astype(np.float64) - fq.astype(np.float64) * float(n)           # the fma is exact on these operands
        q = fq.astype(np.uint32) + (np.abs(fr) >= float(n)).astype(np.uint32)
        r = (y - q * np.uint32(n)) & np.uint32(0xFFFFFF)
        bad = np.flatnonzero(r != y % np.uint32(n))
        assert len(bad) == expected, (n, len(bad))
        if len(bad):
            assert np.all(y[bad] % n == n - 1) and int(y[bad].min()) >= (1 << 22) and np.all(r[bad] == 0xFFFFFF)
