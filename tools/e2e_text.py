#!/usr/bin/env python3
"""End-to-end experiments with the `salt` binary on the GRCh38-scale workload (GPU box): builds the index once, writes one FASTQ file,
then runs `salt -d -c` for every settings string given on the command line ("ENV=VAL,ENV=VAL"; "" = defaults).
A setting IN=bgzf / IN=gz runs on a blocked-gzip / plain-gzip copy of the FASTQ (written once, 32 processes); OUT=null sends the SAM to
/dev/null; PROF=<dir> runs under rocprofv3 --kernel-trace --stats.
usage: tools/e2e_text.py <n_reads> [settings ...]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
from salt_amd import workload
name = os.environ.get("SALT_E2E_WORKLOAD", "grch38")
n = int(sys.argv[1]); settings = sys.argv[2:] or [""]
cache = os.environ.get("SALT_BENCH_CACHE", "/tmp/salt_bench_cache")
dev = torch.device("cuda", 0)
g, p, m = workload.generate_device(name, dev)
w = workload.prepare(name, cache, gpu_device=0, arrays=(g, p, m))
site = workload.make_site_map(g.numel(), p, m)
fq = os.path.join(w["dir"], "e2e.fq")
with open(fq, "wb") as f:
    done = 0
    while done < n:
        k = min(1000000, n - done)
        seqs, _, _, _ = workload.make_reads_hash(g, site, k, 100, seed=77, batch=done // 1000000)
        f.write(workload.fastq_bytes(seqs.cpu().numpy(), k, 100, first_id=done))
        done += k
del g, site
torch.cuda.empty_cache()


def _bgzf_piece(args):
    import struct, zlib
    path, lo, hi = args
    data = open(path, "rb").read()[lo:hi] if False else None
    with open(path, "rb") as f:
        f.seek(lo); data = f.read(hi - lo)
    out = bytearray()
    for o in range(0, len(data), 65280):
        raw = data[o:o + 65280]
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = c.compress(raw) + c.flush()
        out += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1)
        out += body + struct.pack("<II", zlib.crc32(raw) & 0xFFFFFFFF, len(raw))
    return bytes(out)


def make_bgzf(src, dst):
    import multiprocessing as mp
    size = os.path.getsize(src)
    step = 65280 * 256
    with mp.Pool(32) as pool, open(dst, "wb") as f:
        for piece in pool.imap(_bgzf_piece, [(src, lo, min(size, lo + step)) for lo in range(0, size, step)]):
            f.write(piece)
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


made = {}
salt = os.path.join(ROOT, "salt_amd", "bin", "salt")
for s in settings:
    env = dict(os.environ)
    for kv in filter(None, s.split(",")):
        a, b = kv.split("=")
        env[a] = b
    sam = os.path.join(w["dir"], "e2e.sam")
    to_null = env.pop("OUT", "") == "null"
    kind = env.pop("IN", "")
    src = fq
    if kind:
        src = fq + (".bgzf.gz" if kind == "bgzf" else ".plain.gz")
        if kind not in made:
            tz = time.time()
            if kind == "bgzf":
                make_bgzf(fq, src)
            else:
                subprocess.run("gzip -1 -c %s > %s" % (fq, src), shell=True, check=True)
            made[kind] = 1
            print("   (%s written in %.1f s, %.2f GB)" % (src, time.time() - tz, os.path.getsize(src) / 1e9))
    t0 = time.time()
    cmd = [salt, "-d", "-c", "-t", env.pop("T", "64"), w["prefix"], src]
    prof = env.pop("PROF", "")                                  # PROF=<dir>: the run under rocprofv3 --kernel-trace --stats, summary CSV into <dir>
    if prof:
        env["TMPDIR"] = "/tmp"
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.abspath(prof), "-o", "kt", "--"] + cmd
    with open("/dev/null" if to_null else sam, "wb") as fo:
        r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, env=env, cwd="/tmp" if prof else None)
    tail = [l for l in r.stderr.decode().splitlines() if l.startswith("[salt") or l.startswith("[alnse_core]: total")]
    print("== %s  (rc %d, process %.1f s)" % (s or "defaults", r.returncode, time.time() - t0))
    for l in tail:
        print("   ", l)
    sys.stdout.flush()
