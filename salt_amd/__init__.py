"""salt_amd -- MI355X-native implementation of salt's per-read alignment hot path.

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/salt_gpu.h), host/ (index
files + SAM text, include/salt_host.h) and api.py (ctypes bindings that mirror the reference's
batch-level interface)."""
from .api import (AlnOpt, GpuAligner, Index, SaltError, RESULT_DTYPE, read_fastq, sam_text, sam_text_pe, interleave_pairs,  # noqa: F401
                  gpu_lib, host_lib, idx_build, idx_build_mem, suffix_array, IDX_NO_LP)
