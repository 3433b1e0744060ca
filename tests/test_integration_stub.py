"""INTEGRATION.md section B is the patch a maintainer of the reference would apply: its C must compile against the reference's own
headers (query_t, index_t, aln_opt_t, kvec, kstring) and include/salt_gpu.h.  Build-container check (the reference does not travel
to the GPU box): gcc -fsyntax-only -I/root/reference/Align_src."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference/Align_src"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference sources are only present in the build container")
def test_integration_stub_compiles_against_the_reference_headers(tmp_path):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```c\n(.*?)```", text, flags=re.S)
    assert len(blocks) >= 2
    se, pe = blocks[0], blocks[1]
    # the paired-end block elides the packing code it shares with the single-end one ("/* ... */" lines): give it the declarations it uses
    pe = pe.replace("/* ... pack seqs / offs exactly as in alnse_core1_gpu ... */",
                    "uint8_t *seqs = 0; uint32_t *offs = 0; salt_result_t *res = 0;")
    src = tmp_path / "alnse_gpu.c"
    # sam.h pulls in query.h and aln.h (-> indexio.h, which has no include guard: it must come in exactly once)
    src.write_text("#include <stdio.h>\n#include <stdlib.h>\n#include <string.h>\n#include \"sam.h\"\n" + se + "\n" + pe + "\n")
    p = subprocess.run(["gcc", "-std=gnu99", "-fsyntax-only", "-w", "-I", REF, "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
