"""N > 1 path on CPU: world size 2 over gloo.  The GPU batch call is replaced by a stand-in (the CPU oracle,
used here as the checker's twin so that the merged SAM can be compared with the committed golden SAM);
what is under test is salt_amd.dist -- batch sharding, the image broadcast and the ordered merge."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import LAMBDA, ROOT

WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
root = sys.argv[1]; sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import salt_amd
from salt_amd import dist as sd
import oracle_py
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = os.path.join(root, "tests", "golden", "lambda")
# 1. the "index image": rank 0 owns some bytes, everybody must end up with the same bytes
img = torch.from_numpy(np.frombuffer(open(os.path.join(g, "idx.C.bwt"), "rb").read(), dtype=np.uint8).copy()) if rank == 0 else None
img = sd.broadcast_bytes(img, 0)
want = np.frombuffer(open(os.path.join(g, "idx.C.bwt"), "rb").read(), dtype=np.uint8)
assert img.numpy().tobytes() == want.tobytes()
# 2. sharded alignment with a 300-read batch (7 batches over 2 ranks), merged in input order on rank 0
idx = salt_amd.Index.reload(os.path.join(g, "idx"))
opt, _ = salt_amd.AlnOpt.from_argv(["-t", "4"], idx.l_seed)
names, seqs, offs, quals = salt_amd.read_fastq(os.path.join(g, "reads_se.fq"))
ora = oracle_py.Oracle(os.path.join(g, "idx"))
def align_fn(s, o):
    r = ora.align(ora.opt(), s, o)
    out = np.zeros(len(r), dtype=salt_amd.RESULT_DTYPE)          # oracle rows -> product rows (checker's twin)
    for f in ("pos", "n_diff", "is_gap", "mapq", "b0", "b1"): out[f] = r[f]
    out["strand"] = r["strand"].astype(np.uint8); out["seq_end"] = r["seq_end"]; out["n_hits"] = r["n_hits"]
    for f in ("pos", "n_diff", "is_gap", "strand"): out["hits"][f] = r["hits"][f]
    for i in range(len(r)):
        c = r["cigar"][i].decode(); ops = []
        num = ""
        for ch in c:
            if ch.isdigit(): num += ch
            else: ops.append((int(num) << 4) | "MID".index(ch)); num = ""
        out["n_cigar"][i] = len(ops); out["cigar"][i, :len(ops)] = ops
    return out
def format_fn(lo, hi, res):
    return b"".join(idx.samse(opt, names[i], seqs[offs[i]:offs[i + 1]], quals[i], res[i - lo:i - lo + 1]) + b"\n" for i in range(lo, hi))
parts = sd.align_sharded(align_fn, format_fn, seqs, offs, rank, world, batch=300)
assert sorted(parts) == [i for i in range(7) if i % world == rank]
merged = sd.gather_ordered(parts, 7, dst=0)
if rank == 0:
    sam = idx.sam_header(opt) + b"".join(merged)
    open(sys.argv[2], "wb").write(sam)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_shard_broadcast_and_merge(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "merged.sam"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script), ROOT, str(out)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    # (the stand-in carries no XA CIGARs, so the option set without -c is the one compared)
    assert out.read_bytes() == open(os.path.join(LAMBDA, "expect_se_plain_t4.sam"), "rb").read()


def test_batch_sharding_is_a_partition():
    from salt_amd import dist as sd
    for n, world, batch in ((0, 2, 10), (1, 8, 10), (95, 2, 10), (1000, 8, 7), (100000, 3, 100000)):
        seen = []
        for r in range(world):
            seen += [(lo, hi) for _, lo, hi in sd.my_batches(n, r, world, batch)]
        seen.sort()
        assert seen == sd.batch_bounds(n, batch)
        assert sum(hi - lo for lo, hi in seen) == n
