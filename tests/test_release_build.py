"""The release library carries no switch that changes a result: the phase-timing diagnostics that let kernels leave
early (SALT_GPU_HEAVY_STOP, SALT_GPU_LIGHT_STOP, SALT_GPU_SW_SKIP_TB) are compiled only with -DSALT_DIAG (`make -C salt_amd/csrc DIAG=1`,
its own file libsalt_gpu_diag.so); the performance-neutral A/B switches stay."""
import os
import subprocess

from conftest import ROOT


def test_release_library_has_no_result_changing_switches():
    subprocess.run(["make", "-j6", "-C", os.path.join(ROOT, "salt_amd", "csrc")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    blob = open(os.path.join(ROOT, "salt_amd", "lib", "libsalt_gpu.so"), "rb").read()
    for name in (b"SALT_GPU_HEAVY_STOP", b"SALT_GPU_LIGHT_STOP", b"SALT_GPU_SW_SKIP_TB"):
        assert name not in blob, name
    for name in (b"SALT_GPU_LKT_LEN", b"SALT_GPU_NO_CTX", b"SALT_GPU_HEAVY_PER_CU"):      # the neutral ones are still there
        assert name in blob, name
    assert not os.path.exists(os.path.join(ROOT, "salt_amd", "lib", "libsalt_gpu_diag.so")) or \
        b"SALT_GPU_HEAVY_STOP" in open(os.path.join(ROOT, "salt_amd", "lib", "libsalt_gpu_diag.so"), "rb").read()
