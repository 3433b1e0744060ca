#!/usr/bin/env python3
"""bench.py -- throughput of salt's single-end alignment hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload grch38|grch38_tenth|chr21|grch38_mini|mini|tiny]

Workload (default `grch38` = BASELINE.json configs[2] restated as seeded synthetic data): a 3.1e9-base genome in 24
contigs with 14.8 M SNPs, generated on the GPU, indexed on the GPU by the product's own salt-idx (device suffix sorter), k = 21;
100-base single-end reads.  A "step" is one pass of the hot path (k_pack ... k_cigar behind salt_gpu_align_se_resident) over one
batch of 1 000 000 reads that is already resident in HBM.  --batches (8) DISTINCT batches are resident and the steps rotate
through them, dealt round-robin to --streams (4) workspaces, each on its own HIP stream, as `salt` drives a GPU with several align
workers.  Per-GPU work is fixed (weak scaling): every rank aligns its own read shards against its own replica of the device
index; rank 0 builds + packs the index and the other ranks receive its compact part by one RCCL broadcast (no collective on the
data path).

Prints ONE JSON line (rank 0) with the metric of BASELINE.json plus
  "stages_s":     what the run spent before the timed region (generate / index / load / attach / broadcast / reads)
  "roofline":     the dominant single kernel: device-layout algorithmic bytes per launch (kernel counters) / its HIP-event time
                  vs 8 TB/s, PMC traffic and the issue-rate bound from profiles/ when they hold this workload
  "cpu_baseline": the CPU oracle (bit-exact restatement of the reference) on a bounded sample of the same reads on this box's
                  host cores, also used to check the GPU results ("parity")
  "e2e":          the `salt` binary, FASTQ text in -> SAM text out on reads of the same workload: SURVEY 8d's wall-clock metric.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
RANDOM_REQ_GPS = 48.0          # measured: random 64-byte requests per second (in G) the memory system serves from tables of 8-128 GiB,
                               # tools/ubench/gather (profiles/r02/gather_rate.txt: 38-48 G/s, i.e. 2.4-3.1 TB/s of sectors; 55 G/s from 1-2 GiB)
SECTOR = 64                    # bytes a random access moves at least (one L2 / fabric sector)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def device_bytes(ctr, n_reads, L, spr):
    """Device-layout algorithmic bytes of ONE launch per kernel, from the kernels' own access counters (salt_gpu_ws_counters,
    DESIGN 5): what the device structures must move -- one 64-byte sector per W-mer gather, the 32-byte C / 64-byte R Occ blocks
    actually fetched, 4 bytes per suffix-array / R-position load (16 when the row comes from the context table), 8 per text word pair,
    16 per verify lane-load (64 per candidate window), the packed read records, seed intervals and result rows each kernel reads and writes."""
    items = n_reads * 2 * spr
    nw8, nw16, nw32 = (L + 7) // 8, (L + 15) // 16, (L + 31) // 32
    pm = ((2 * nw8 + 1 + 3) & ~3) * 4
    tb = ((2 * nw16 + 2 * nw32 + 1 + 3) & ~3) * 4
    b = {}
    b["k_pack"] = n_reads * (L + 4 + pm + tb)
    b["k_seed"] = (n_reads * tb + ctr["d_wlkt"] * SECTOR + ctr["d_cocc_seed"] * 32 + ctr["d_rocc_seed"] * 64 + ctr["d_sa_seed"] * 4
                   + ctr["d_text_seed"] * 8 + items * 32)
    b["k_light"] = (n_reads * pm + items * 32 + ctr["d_sa_light"] * 4 + ctr["d_verify_light"] * 16 + ctr["d_out_light"])
    b["k_heavy"] = (ctr["heavy_reads"] * (pm + 2 * spr * 32) + ctr["d_sa_heavy"] * 4 + ctr.get("d_ctx_rows", 0) * 12 + ctr["d_verify_heavy"] * 16 + ctr["d_out_heavy"])
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default=os.environ.get("SALT_BENCH_WORKLOAD", "grch38"))
    ap.add_argument("--batches", type=int, default=int(os.environ.get("SALT_BENCH_BATCHES", "8")), help="distinct resident read batches the steps rotate through")
    ap.add_argument("--cpu-sample", type=int, default=1000000, help="reads given to the CPU baseline / parity check")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--e2e-reads", type=int, default=32000000, help="reads of the end-to-end leg (`salt` binary, FASTQ -> SAM); 0 = skip")
    ap.add_argument("--e2e-pairs", type=int, default=4000000, help="pairs (2 x 150) of the paired-end end-to-end leg (`salt -p`); 0 = skip")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU per step (experiments; default: the workload's own batch)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("SALT_BENCH_STREAMS", "4")),
                    help="workspaces / HIP streams per GPU the steps are dealt to round-robin (salt runs 2-4 align workers per GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    import salt_amd
    from salt_amd import workload

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    # rehearsal of the N > 1 path on a one-GPU box: SALT_BENCH_SAME_GPU=1 puts every rank on GPU 0 and SALT_BENCH_BACKEND=gloo
    # replaces RCCL (which refuses two ranks on one device); the driver's multi-GPU runs use neither
    if os.environ.get("SALT_BENCH_SAME_GPU"):
        local_rank = 0
    backend = os.environ.get("SALT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    cfg = workload.CONFIGS[args.workload]
    L, n_reads = cfg["read_len"], (args.reads or cfg["n_reads"])
    cache = os.environ.get("SALT_BENCH_CACHE", "/tmp/salt_bench_cache")
    stages = {}

    # ---- genome + SNPs: every rank generates them on its own GPU (it draws its read shards from them) ----
    t0 = time.time()
    genome, pos, mask = workload.generate_device(args.workload, dev)
    site = workload.make_site_map(genome.numel(), pos, mask)
    torch.cuda.synchronize()
    stages["generate"] = round(time.time() - t0, 2)

    # ---- index files: rank 0 builds them with the product's own salt-idx (suffixes sorted on its GPU) ----
    w = None
    if rank == 0:
        w = workload.prepare(args.workload, cache, gpu_device=local_rank, log=log, arrays=(genome, pos, mask))
        stages.update(w["times"])
    if world > 1:
        dist.barrier()

    # ---- device index: rank 0 packs, the others get the compact image by one broadcast over RCCL ----
    idx = None
    nbytes = cbytes = 0
    if rank == 0:
        t0 = time.time()
        idx = salt_amd.Index.reload(w["prefix"], rebuild_lkt=False)
        stages["load_files"] = round(time.time() - t0, 2)
        torch.cuda.empty_cache()                             # the generators' temporaries: the device image sizes its k-mer table by what is free
        t0 = time.time()
        aln = salt_amd.GpuAligner(idx, device=local_rank, max_reads=n_reads, max_bases=n_reads * L)
        torch.cuda.synchronize()
        stages["attach"] = round(time.time() - t0, 2)
        nbytes = aln.image()[1]
        cbytes = aln.image_compact()[1]
    if world > 1:
        # only the compact part travels (FM-indexes, suffix arrays, mixRef, 2-bit text); the 16-byte W-mer table that
        # ends the image is a function of it and each rank tabulates its own copy
        t0 = time.time()
        sz = torch.tensor([cbytes], dtype=torch.int64, device=dev)
        dist.broadcast(sz, 0)
        cbytes = int(sz.item())
        img = torch.empty(cbytes, dtype=torch.uint8, device=dev)
        if rank == 0:
            aln.image_copy(img.data_ptr(), cbytes)
        chunk = 1 << 30
        for o in range(0, cbytes, chunk):
            dist.broadcast(img[o:o + chunk], 0)
        if rank != 0:
            aln = salt_amd.GpuAligner(None, device=local_rank, max_reads=n_reads, max_bases=n_reads * L,
                                      compact=(img.data_ptr(), cbytes))
        torch.cuda.synchronize()
        del img
        stages["broadcast_and_attach_replicas"] = round(time.time() - t0, 2)
    torch.cuda.synchronize()
    if rank == 0:
        log("device index: %.2f GiB (compact part, the only thing broadcast: %.2f GiB); stages so far: %s" % (nbytes / 2**30, cbytes / 2**30, stages))

    # ---- distinct read batches, resident ----
    t0 = time.time()
    n_batches = max(1, args.batches)
    batches = []
    for b in range(n_batches):
        seqs, offs, _, _ = workload.make_reads_hash(genome, site, n_reads, L, seed=1 + rank, batch=b)
        batches.append((seqs, offs))
    torch.cuda.synchronize()
    stages["reads"] = round(time.time() - t0, 2)

    opt = salt_amd.AlnOpt(l_seed=cfg["k"])
    spr = (L - cfg["k"]) // cfg["k"] + 1
    # Steps are dealt round-robin to n_streams workspaces, each on its own HIP stream, all on the one device index: a step is
    # still one pass of the whole path over one batch, but the persistent kernels' tails of one batch overlap the wide kernels
    # of the next -- the way `salt` drives a GPU with 2-4 align workers (salt_main.cc).
    n_streams = max(1, min(args.streams, max(args.steps, 1)))
    alns = [aln] + [aln.fork() for _ in range(n_streams - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    d_ress = [torch.zeros(n_reads * salt_amd.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in range(n_streams)]

    def step(i, k=None):
        k = i % n_streams if k is None else k
        s, o = batches[i % n_batches]
        alns[k].align_resident(opt, n_reads, L, s.data_ptr(), o.data_ptr(), d_ress[k].data_ptr(), streams[k].cuda_stream)

    torch.cuda.synchronize()
    for i in range(max(args.warmup, 0)):
        step(i)
    torch.cuda.synchronize()
    # serialized steps first (one workspace, one stream, every batch once): per-kernel durations of a kernel that has the GPU to
    # itself -- what a roofline fraction is about; in the timed region a kernel shares the GPU with the other batches' kernels
    for a in alns:
        a.timing(True)
    n_serial = max(3, n_batches)
    for i in range(n_serial):
        step(i, 0)
    torch.cuda.synchronize()
    serial_kms, serial_calls = aln.kernel_ms()
    serial_kms = {k: v / max(serial_calls, 1) for k, v in serial_kms.items()}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kms, n_calls = {}, 0                                   # live: HIP events over the timed region, all streams
    for a in alns:
        km, nc = a.kernel_ms()
        n_calls += nc
        for k, v in km.items():
            kms[k] = kms.get(k, 0.0) + v
    kms = {k: v / max(n_calls, 1) for k, v in kms.items()}
    for a in alns:
        a.timing(False)

    out = None
    if rank == 0:
        total_reads = n_reads * world * args.steps
        value = total_reads / dt / 1e6
        names = {"grch38": "GRCh38-scale synthetic (SURVEY 8d config 3 = BASELINE configs[2])", "chr21": "chr21-scale synthetic (SURVEY 8d config 2)"}
        out = {
            "metric": "Mreads/s aligned (100 bp SE, GRCh38+snp144) at 1/2/4/8 MI355X vs CPU ref",
            "value": round(value, 4), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "%s: %d bp genome in %d contigs, %d SNPs, k=%d, %d x %d bp SE reads per GPU per step, %d distinct resident batches "
                                   "rotated, inputs and results resident in HBM" % (names.get(args.workload, args.workload + " synthetic"), cfg["genome_len"],
                                                                                    cfg.get("contigs", 1), cfg["n_snps"], cfg["k"], n_reads, L, n_batches),
                       "reads_per_gpu_per_step": n_reads, "read_len": L, "distinct_batches": n_batches, "options": "default (-s 50 -m 1000, overlap = k)",
                       "parallelism": "reads sharded over %d GPU(s), index replicated (one RCCL broadcast of its compact part); "
                                      "steps dealt round-robin to %d workspace(s)/HIP stream(s) per GPU" % (world, n_streams),
                       "streams_per_gpu": n_streams},
            "stages_s": stages,
            "kernel_ms": {k: round(v, 3) for k, v in kms.items()},
            "kernel_ms_serialized": {k: round(v, 3) for k, v in serial_kms.items()},
        }
        # ---- one more serialized step of batch 0 with the access counters on (k_light instead of k_light2: same accesses) ----
        copt = salt_amd.AlnOpt(l_seed=cfg["k"], collect_counters=1)
        aln.counters()
        s0, o0 = batches[0]
        aln.align_resident(copt, n_reads, L, s0.data_ptr(), o0.data_ptr(), d_ress[0].data_ptr(), streams[0].cuda_stream)
        torch.cuda.synchronize()
        ctr = aln.counters()
        qc = aln.queue_counts()
        out["queue_counts"] = dict(zip(("heavy_reads", "gapped_reads", "k_gap_items", "k_cigar_items"), [qc[0], qc[2], qc[5], qc[6]]))
        ctr["heavy_reads"] = qc[0]
        try:
            db = device_bytes(ctr, n_reads, L, spr)
        except KeyError:
            db = None
        dom = max(serial_kms, key=lambda k: serial_kms[k])
        prof = {}
        pf = os.path.join(ROOT, "profiles", "r02", "pmc_summary_%s.json" % args.workload)
        if os.path.exists(pf):
            try:
                prof = json.load(open(pf))
            except Exception:
                prof = {}
        roof = {"bound": "hbm", "kernel": dom, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "timing": "HIP events on the launch stream over %d serialized steps of this run, every resident batch once (kernel_ms_serialized)" % n_serial}
        if db and dom in db:
            ach = db[dom] / (serial_kms[dom] / 1e3) / 1e9
            roof.update({"achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "bytes_per_launch": int(db[dom]),
                         "avg_launch_ms": round(serial_kms[dom], 3),
                         "all_kernels": {k: {"bytes_per_launch": int(v), "ms": round(serial_kms.get(k, 0.0), 3),
                                             "GBps": round(v / (serial_kms[k] / 1e3) / 1e9, 1) if serial_kms.get(k) else None} for k, v in db.items()},
                         "whole_step_GBps_wall": round(sum(db.values()) / (dt / args.steps) / 1e9, 1),
                         "bytes_model": "device layout: 64 B per W-mer gather, 32/64 B per C/R Occ block fetched, 4 B per SA / R-position load, 8 B per text "
                                        "word pair, 16 B per verify lane-load, packed read records, 32 B per seed interval pair, result rows; counted by the kernels"})
        tr = prof.get("hbm_bytes_per_launch", {}).get(dom)
        roof["traffic"] = tr
        pk = prof.get("per_kernel_mean", {})
        if tr:
            roof["traffic_GBps"] = round(tr / (serial_kms[dom] / 1e3) / 1e9, 1)
            roof["traffic_note"] = ("FETCH_SIZE + WRITE_SIZE of profiles/r02 (separate --pmc passes), raw: tools/ubench/gather shows FETCH_SIZE = 64.0 B per random "
                                    "4/16/32/64-byte record, i.e. exact for these gather shapes (no x2 streaming correction applies)")
        rq = pk.get(dom, {}).get("TCC_EA0_RDREQ_sum")
        if rq:
            # how far the kernel is from the memory system's limit for its access shape: random 64-byte requests per second (a second adjacent sector per window is free)
            roof["random_requests"] = {"per_launch": int(rq), "G_per_s": round(rq / (serial_kms[dom] / 1e3) / 1e9, 1), "ceiling_G_per_s": RANDOM_REQ_GPS,
                                       "frac_of_ceiling": round(rq / (serial_kms[dom] / 1e3) / 1e9 / RANDOM_REQ_GPS, 3),
                                       "whole_step_G_per_s_wall": round(sum(v.get("TCC_EA0_RDREQ_sum", 0) for v in pk.values()) / (dt / args.steps) / 1e9, 1),
                                       "source": "TCC_EA0_RDREQ_sum per launch (profiles/r02/pmc_tcc_*.csv) / this run's HIP-event time; ceiling: tools/ubench/gather"}
        alltr = prof.get("hbm_bytes_per_launch", {})
        if alltr:
            roof["whole_step_traffic_GBps_wall"] = round(sum(alltr.values()) / (dt / args.steps) / 1e9, 1)
        ir = prof.get("issue", {}).get(dom)
        if ir:
            roof["issue_bound"] = ir
        roof["limiter"] = ("memory latency x resident waves, not a bandwidth: k_seed and k_light2 fill every wave slot (8 per SIMD) with dependent chains of random loads "
                           "(a W-mer gather, then up to k - W Occ steps per seed; 3 round trips per read pair), k_heavy's reads make ~20 dependent round trips each on 6 one-wave "
                           "blocks per CU.  With 4 or more streams the step is pinned at the sum of the slot-filling kernels plus the part of k_heavy that does not hide behind them "
                           "(2 streams: the serialized sum; 4, 6, 8: the same plateau).  random_requests.frac_of_ceiling and frac (bytes) say how far the memory system is from ITS "
                           "limits: about half of the random-request rate and a fifth of the streaming peak.  Halving k_heavy's requests (the context table, DESIGN 3) "
                           "took 14 % off its time: that is what a latency bound looks like")
        roof["counters"] = {k: int(v) for k, v in ctr.items() if k.startswith("d_")}
        out["roofline"] = roof

        # ---- CPU baseline + parity check on a bounded sample (oracle = checker, never the product) ----
        if not args.no_cpu:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_py
            ns = min(args.cpu_sample, n_reads)
            t1 = time.time()
            ora = oracle_py.Oracle(w["prefix"])
            stages["oracle_load_files"] = round(time.time() - t1, 2)
            oo = ora.opt()
            cores = min(os.cpu_count() or 1, 64)
            hs, ho = s0[:ns * L].cpu().numpy(), o0[:ns + 1].cpu().numpy().view(np.uint32)
            t1 = time.perf_counter()
            ores = ora.align(oo, hs, ho, n_threads=cores)
            cpu_dt = time.perf_counter() - t1
            gres = d_ress[0].cpu().numpy().view(salt_amd.RESULT_DTYPE)[:ns]     # batch 0 was the last batch aligned into d_ress[0]
            bad = oracle_py.compare(gres, ores)
            out["cpu_baseline"] = {"value": round(ns / cpu_dt / 1e6, 5), "unit": "Mreads/s", "cores": cores, "kind": "port",
                                   "sample": "first %d reads of batch 0 of the same workload, oracle/libsalt_oracle.so (bit-exact CPU restatement "
                                             "of the reference), %d threads, align time only (%.1f s)" % (ns, cores, cpu_dt),
                                   "speedup_1gpu": round(value / world / (ns / cpu_dt / 1e6), 1)}
            out["parity"] = {"checked_reads": int(ns), "mismatching_reads": int(len(bad)),
                             "mapped_fraction": round(float((gres["pos"] != 0xFFFFFFFF).mean()), 5)}
            ora.close()
    # the index image and the workspaces go before the end-to-end legs: the `salt` processes attach their own image, and two of them do
    # not fit one GPU (a salt that finds 70 GiB free attaches a narrower k-mer table and no context table)
    for a in alns[1:]:
        a.close()
    aln.close()
    if idx is not None:
        idx.destroy()
    if rank == 0:
        del d_ress, batches
        torch.cuda.empty_cache()
        # ---- end to end: the drop-in binary, FASTQ text -> SAM text, wall clock (SURVEY 8d's metric; never `value`) ----
        if args.e2e_reads > 0:
            try:
                out["e2e"] = e2e_leg(args, cfg, w, genome, site, workload, torch, np, log)
            except Exception as ex:                                    # the leg is a report, not a gate
                out["e2e"] = {"error": str(ex)[:300]}
        if args.e2e_pairs > 0 and args.e2e_reads > 0:
            try:
                out["e2e_pe"] = e2e_pe_leg(args, w, genome, site, workload, torch, np, log)
            except Exception as ex:
                out["e2e_pe"] = {"error": str(ex)[:300]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def e2e_leg(args, cfg, w, genome, site, workload, torch, np, log):
    """`salt -d -c` on a FASTQ file of the same workload: the binary's own clock (starts when the index is loaded and attached, like the
    reference's; ends with the last SAM byte written), plus the whole process."""
    L = cfg["read_len"]
    d = w["dir"]
    fq = os.path.join(d, "e2e.fq")
    n = args.e2e_reads
    t0 = time.time()
    with open(fq, "wb") as f:
        done = 0
        while done < n:
            m = min(1000000, n - done)
            seqs, _, _, _ = workload.make_reads_hash(genome, site, m, L, seed=77, batch=done // 1000000)
            f.write(workload.fastq_bytes(seqs.cpu().numpy(), m, L, first_id=done))
            done += m
    t_write = time.time() - t0
    threads = min(os.cpu_count() or 1, 64)
    salt = os.path.join(ROOT, "salt_amd", "bin", "salt")
    sam = os.path.join(d, "e2e.sam")
    t0 = time.time()
    with open(sam, "wb") as fo:
        p = subprocess.run([salt, "-d", "-c", "-t", str(threads), w["prefix"], fq], stdout=fo, stderr=subprocess.PIPE, timeout=900)
    wall = time.time() - t0
    err = p.stderr.decode(errors="replace")
    if p.returncode != 0:
        raise RuntimeError("salt exited %d: %s" % (p.returncode, " | ".join(l for l in err.splitlines() if "have been aligned" not in l)[-280:]))
    align_s, detail = None, None
    for line in err.splitlines():
        if line.startswith("[alnse_core]: total"):
            align_s = float(line.split()[2])
        if line.startswith("[salt] text path:") or line.startswith("[salt] host phases"):
            detail = line[7:]
    sam_bytes = os.path.getsize(sam)
    os.unlink(sam); os.unlink(fq)
    return {"value": round(n / align_s / 1e6, 3) if align_s else None, "unit": "Mreads/s", "reads": n, "threads": threads,
            "what": "salt -d -c -t %d <idx> reads.fq > out.sam: FASTQ text in, SAM text out (%.2f GB), PCIe and host I/O included; clock = the binary's "
                    "[alnse_core] total (restarted where the reference restarts its own, behind the index reload, alnse.c:1366; workspace set-up included; ends with the "
                    "last SAM byte written; index load + attach excluded, as SURVEY 8d defines it)" % (threads, sam_bytes / 1e9),
            "align_wall_s": align_s, "process_wall_s": round(wall, 2), "fastq_write_s": round(t_write, 2), "pipeline": detail}


def e2e_pe_leg(args, w, genome, site, workload, torch, np, log):
    """`salt -d -c -p -a 250 -b 550` on two FASTQ files of 2 x 150-base pairs of the same genome (BASELINE's paired-end configuration): the
    binary's own clock as in e2e_leg."""
    L, n, d = 150, args.e2e_pairs, w["dir"]
    fq = [os.path.join(d, "e2e_1.fq"), os.path.join(d, "e2e_2.fq")]
    with open(fq[0], "wb") as f1, open(fq[1], "wb") as f2:
        done = 0
        while done < n:
            k = min(500000, n - done)
            seqs, _, _, _, _ = workload.make_pairs_hash(genome, site, k, L, seed=78, batch=done // 500000)
            r = seqs.view(k, 2, L).cpu().numpy()
            f1.write(workload.fastq_bytes(r[:, 0, :].reshape(-1), k, L, first_id=done))
            f2.write(workload.fastq_bytes(r[:, 1, :].reshape(-1), k, L, first_id=done))
            done += k
    threads = min(os.cpu_count() or 1, 64)
    sam = os.path.join(d, "e2e_pe.sam")
    t0 = time.time()
    with open(sam, "wb") as fo:
        p = subprocess.run([os.path.join(ROOT, "salt_amd", "bin", "salt"), "-d", "-c", "-p", "-a", "250", "-b", "550", "-t", str(threads), w["prefix"]] + fq,
                           stdout=fo, stderr=subprocess.PIPE, timeout=900)
    wall = time.time() - t0
    err = p.stderr.decode(errors="replace")
    if p.returncode != 0:
        raise RuntimeError("salt -p exited %d: %s" % (p.returncode, " | ".join(l for l in err.splitlines() if "have been aligned" not in l)[-280:]))
    align_s, detail = None, None
    for line in err.splitlines():
        if line.startswith("[alnpe_core]: total") or line.startswith("[alnse_core]: total"):
            align_s = float(line.split()[2])
        if line.startswith("[salt] text path") or line.startswith("[salt] host phases"):
            detail = line[7:]
    sam_bytes = os.path.getsize(sam)
    for f in fq + [sam]:
        os.unlink(f)
    return {"value": round(2 * n / align_s / 1e6, 3) if align_s else None, "unit": "M mates/s", "pairs": n, "read_len": L, "threads": threads,
            "what": "salt -d -c -p -a 250 -b 550 -t %d <idx> r1.fq r2.fq > out.sam (%.2f GB of SAM); the binary's clock as in e2e" % (threads, sam_bytes / 1e9),
            "align_wall_s": align_s, "process_wall_s": round(wall, 2), "pipeline": detail}


if __name__ == "__main__":
    main()
