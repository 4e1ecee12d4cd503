"""Index algebra of the strip Legendre kernel (ace_amd/csrc/strip.hip) on the CPU: tests/emul/strip_emul.cpp restates the
kernel's lane / fragment / tile / store mapping on top of the real operand packer (ace_amd/csrc/strip_pack.h) and
compares with direct sums, forward and inverse, ragged and full shapes (incl. 180 x 180 x 181)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_strip_kernel_index_algebra(tmp_path):
    exe = str(tmp_path / "strip_emul")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "emul", "strip_emul.cpp")], check=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "worst" in res.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_folded_strip_kernel_index_algebra(tmp_path):
    """ace_amd/csrc/strip_fold.hip (equatorially folded Legendre stages): geometry, packer, the two range-checked strip
    descriptors, unit / pair loop, row maps and store masks against UNFOLDED direct sums - even and odd nlat, ragged shapes,
    every output element written exactly once."""
    exe = str(tmp_path / "strip_fold_emul")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "emul", "strip_fold_emul.cpp")], check=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "worst" in res.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_conv_ws_work_decomposition(tmp_path):
    """ace_amd/csrc/ws_plan.h (shared by the kernel and its launcher): every (channel slice, pixel tile) unit exactly once,
    every (statistics slot, row) exactly one partial, over a sweep of channel counts and field sizes."""
    exe = str(tmp_path / "ws_plan_emul")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "emul", "ws_plan_emul.cpp")], check=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "shapes ok" in res.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_small_fft_templates(tmp_path):
    """ace_amd/csrc/small_fft.h (the compile-time mixed-radix FFTs of the longitude transform, fft.hip) compiled for the host:
    complex forward / inverse and real-input half spectra against direct double-precision sums, every level length in use."""
    exe = str(tmp_path / "small_fft_emul")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "emul", "small_fft_emul.cpp")], check=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "small ffts ok" in res.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_conditional_layer_norm_mfma_decomposition(tmp_path):
    """ace_amd/csrc/cln_mfma.hip (single-pass conditional layer norm): waves x row tiles, the A fragments pack_cln_frags builds at
    weight upload, the B fragments built from the conditioning field, the MFMA lane layout and the apply, against the direct
    fp64 formula for C in {256, 512, 768, 1024} and ragged J."""
    exe = str(tmp_path / "cln_mfma_emul")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "emul", "cln_mfma_emul.cpp")], check=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "worst" in res.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_dhconv_work_list(tmp_path):
    """ace_amd/csrc/dhconv_units.h (the launch order of dhconv_strip.hip's workgroups): every row of every (degree, column group)
    in exactly one unit on the owning XCD, chunks of a group neighbours, the four shader engines of an XCD evenly loaded."""
    exe = str(tmp_path / "dhconv_units_emul")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "emul", "dhconv_units_emul.cpp")], check=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "worst" in res.stdout
