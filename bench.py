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
  "pe":           BASELINE.json configs[3] (2 x 150-base pairs, -p -a 250 -b 550, on the same index): the kernel stage of
                  salt_gpu_align_pe_resident on resident batches, per-kernel times, the roofline of its dominant kernel, parity against the
                  oracle on a sample of a timed step's pairs, and the oracle's own rate on this box.
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


def cpu_info():
    """(threads this process may run on, model name) of the box the CPU baseline is timed on."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return n, model


def spread(step_ends_ms):
    """min / median / max of the intervals between consecutive step completions (ms): with several batches in flight a step has no wall
    time of its own, but the completion times of the steps do -- their spacing is the step rate as it varies inside the timed region."""
    t = sorted(step_ends_ms)
    d = [t[0]] + [b - a for a, b in zip(t, t[1:])]
    d.sort()
    return {"min": round(d[0], 3), "median": round(d[len(d) // 2], 3), "max": round(d[-1], 3), "what": "intervals between consecutive step completions (HIP events on each step's stream)"}


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
    ap.add_argument("--e2e-pairs", type=int, default=8000000, help="pairs (2 x 150) of the paired-end end-to-end leg (`salt -p`); 0 = skip")
    ap.add_argument("--pe-pairs", type=int, default=500000, help="pairs (2 x 150) per step of the paired-end kernel-stage leg; 0 = skip")
    ap.add_argument("--pe-steps", type=int, default=16)
    ap.add_argument("--pe-batches", type=int, default=4)
    ap.add_argument("--pe-check", type=int, default=20000, help="pairs of a timed step compared with the oracle (and timed there)")
    ap.add_argument("--mode", choices=("both", "se", "pe"), default="both", help="profiling runs: `pe` = the paired-end leg with a token single-end step, `se` = no paired-end leg")
    ap.add_argument("--no-counters", action="store_true", help="profiling runs: skip the steps that run with the access counters on (they use other kernel variants)")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU per step (experiments; default: the workload's own batch)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("SALT_BENCH_STREAMS", "4")),
                    help="workspaces / HIP streams per GPU the steps are dealt to round-robin (salt runs 2-4 align workers per GPU)")
    args = ap.parse_args()

    if args.mode == "pe":
        args.steps, args.warmup, args.batches, args.e2e_reads, args.e2e_pairs = max(1, args.streams), 1, 1, 0, 0
    if args.mode == "se":
        args.pe_pairs = 0
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
        import datetime
        wait = datetime.timedelta(minutes=40)                               # ranks > 0 sit in the last barrier while rank 0 runs the end-to-end legs (four `salt` runs)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=wait)
        else:
            dist.init_process_group(backend, timeout=wait)

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
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_ev[0].record(streams[0])
    for i in range(args.steps):
        step(i)
        step_ev[i + 1].record(streams[i % n_streams])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the rows the LAST timed step left behind (its batch, its workspace): what the parity leg compares with the oracle
    last = args.steps - 1
    timed_rows = d_ress[last % n_streams].clone() if args.steps > 0 else None
    timed_batch = last % n_batches
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
            "step_ms": spread([step_ev[0].elapsed_time(e) for e in step_ev[1:]]) if args.steps > 0 else None,
            "kernel_ms": {k: round(v, 3) for k, v in kms.items() if v},
            "kernel_ms_serialized": {k: round(v, 3) for k, v in serial_kms.items() if v},
        }
        # ---- one more serialized step of batch 0 with the access counters on (k_light instead of k_light2: same accesses; its rows are NOT
        # the ones compared with the oracle -- those are timed_rows, taken above) ----
        copt = salt_amd.AlnOpt(l_seed=cfg["k"], collect_counters=0 if args.no_counters else 1)
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
        # the dominant kernel among those whose bytes are counted (the memory-bound ones; on a toy workload a gapped-pass kernel can be the longest)
        dom = max([k for k in serial_kms if db and k in db] or list(serial_kms), key=lambda k: serial_kms[k])
        prof, prof_dir = {}, None
        for rd in ("r03", "r02"):                               # the newest committed profile of this workload
            pf = os.path.join(ROOT, "profiles", rd, "pmc_summary_%s.json" % args.workload)
            if os.path.exists(pf):
                try:
                    prof, prof_dir = json.load(open(pf)), "profiles/" + rd
                    break
                except Exception:
                    prof = {}
        pkey = lambda k: "k_light2" if k == "k_light" and "k_light2" in prof.get("per_kernel_mean", {}) else k    # the timed steps run k_light2
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
        # k_seed's timing slot is two kernels (k_seed: table gather + the seeds it finishes itself; k_seed_walk: the queued walks): when it
        # is the dominant one every profile figure below is the sum over both, and the rocprofv3 durations to compare with are both rows
        parts = lambda k: ("k_seed", "k_seed_walk") if k == "k_seed" else (pkey(k),)
        if dom == "k_seed":
            roof["kernel"] = "k_seed (k_seed + k_seed_walk: one timing slot, back to back on the stream)"
        tr = sum(prof.get("hbm_bytes_per_launch", {}).get(q, 0) for q in parts(dom)) or None
        roof["traffic"] = tr
        pk0 = prof.get("per_kernel_mean", {})
        pk = dict(pk0)
        if "k_seed" in pk0 and "k_seed_walk" in pk0:                    # (per-kernel counters of the slot added; issue fractions are not additive and stay per kernel)
            pk["k_seed"] = {c: pk0["k_seed"].get(c, 0.0) + pk0["k_seed_walk"].get(c, 0.0) for c in set(pk0["k_seed"]) | set(pk0["k_seed_walk"])}
        if tr:
            roof["traffic_GBps"] = round(tr / (serial_kms[dom] / 1e3) / 1e9, 1)
            roof["traffic_note"] = ("FETCH_SIZE + WRITE_SIZE of %s (separate --pmc passes)" % prof_dir + ", raw: tools/ubench/gather shows FETCH_SIZE = 64.0 B per random "
                                    "4/16/32/64-byte record, i.e. exact for these gather shapes (no x2 streaming correction applies)")
        rq = pk.get(pkey(dom), {}).get("TCC_EA0_RDREQ_sum")
        if rq:
            # how far the kernel is from the memory system's limit for its access shape: random 64-byte requests per second (a second adjacent sector per window is free)
            def req_rates(k, ms):                                        # read requests (TCC_EA0_RDREQ) + write requests (WRITE_SIZE, KiB, / 64 B) of a kernel
                v = pk.get(pkey(k), {})
                rd, wr = v.get("TCC_EA0_RDREQ_sum", 0.0), v.get("WRITE_SIZE", 0.0) * 1024.0 / SECTOR
                return {"read_requests": int(rd), "write_requests": int(wr), "G_per_s": round((rd + wr) / (ms / 1e3) / 1e9, 1) if ms else None}
            wr = pk.get(pkey(dom), {}).get("WRITE_SIZE", 0.0) * 1024.0 / SECTOR
            roof["random_requests"] = {"per_launch": int(rq), "G_per_s": round(rq / (serial_kms[dom] / 1e3) / 1e9, 1), "ceiling_G_per_s": RANDOM_REQ_GPS,
                                       "frac_of_ceiling": round(rq / (serial_kms[dom] / 1e3) / 1e9 / RANDOM_REQ_GPS, 3),
                                       "with_writes": {"per_launch": int(rq + wr), "G_per_s": round((rq + wr) / (serial_kms[dom] / 1e3) / 1e9, 1),
                                                       "ms_at_the_ceiling": round((rq + wr) / (RANDOM_REQ_GPS * 1e9) * 1e3, 3)},
                                       "all_kernels": {k: req_rates(k, serial_kms.get(k, 0.0)) for k in ("k_seed", "k_light", "k_heavy", "k_gap") if serial_kms.get(k)},
                                       "ceilings_by_shape_G_per_s": {"4-byte": 48, "16-byte": 38, "32-byte": 38, "64-byte (4 lanes x 16 B)": 47,
                                                                     "source": "tools/ubench/gather, the same from 16 ... 240 GiB tables and from 65 000 ... 2 000 000 loads in flight (profiles/r03/gather_rate*.log)"},
                                       "whole_step_G_per_s_wall": round(sum(v.get("TCC_EA0_RDREQ_sum", 0) for v in pk0.values()) / (dt / args.steps) / 1e9, 1),
                                       "source": "TCC_EA0_RDREQ_sum (+ WRITE_SIZE / 64 B) per launch (%s/pmc_tcc_*.csv, pmc_write_*.csv) / this run's HIP-event time; ceiling: tools/ubench/gather; "
                                                 "k_seed in all_kernels is k_seed + k_seed_walk (one timing slot: their counters are added)" % prof_dir}
        alltr = prof.get("hbm_bytes_per_launch", {})
        if alltr:
            roof["whole_step_traffic_GBps_wall"] = round(sum(alltr.values()) / (dt / args.steps) / 1e9, 1)
        ir = {q: prof.get("issue", {}).get(q) for q in parts(dom) if prof.get("issue", {}).get(q)}
        if ir:
            roof["issue_bound"] = ir if len(ir) > 1 else list(ir.values())[0]
        roof["limiter"] = ("the rate at which the memory system serves sector-sized random requests into a 195 GiB image, not its byte bandwidth (random_requests): with the write "
                           "requests counted k_heavy now makes its requests at ~42 G/s and k_seed + k_seed_walk at ~34 G/s, against the 38 - 48 G/s tools/ubench/gather reaches for "
                           "their shapes.  Until this round k_heavy stood at 26 G/s whatever its occupancy or its code: its waves took their reads through ONE atomic counter, which "
                           "serves a pop every ~14 ns (time = 0.08 ms + 14.4 ns x queued reads from 9 000 to 150 000 reads, profiles/r03/heavy_vs_batch.log); with 64 ranged heads the "
                           "kernel follows its waves again (1.165 -> 0.73 ms at 16 blocks per CU, profiles/r03/ab_heavy_ranged_pops.log; DESIGN 5.0).  "
                           "random_requests.with_writes.ms_at_the_ceiling is what the dominant slot's requests would take at the gather ceiling; what moves the step from here is "
                           "fewer requests per read (the context tables did that: DESIGN 3, 4.2, 5.0)")
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
            cores, cpu_model = cpu_info()                                   # every hardware thread this process may use
            sb, ob = batches[timed_batch]
            hs, ho = sb[:ns * L].cpu().numpy(), ob[:ns + 1].cpu().numpy().view(np.uint32)
            t1 = time.perf_counter()
            ores = ora.align(oo, hs, ho, n_threads=cores)
            cpu_dt = time.perf_counter() - t1
            gres = timed_rows.cpu().numpy().view(salt_amd.RESULT_DTYPE)[:ns]   # the rows of the last TIMED step (k_light2 and all), not of the counters step
            bad = oracle_py.compare(gres, ores)
            out["cpu_baseline"] = {"value": round(ns / cpu_dt / 1e6, 5), "unit": "Mreads/s", "cores": cores, "cpu_model": cpu_model, "kind": "port",
                                   "sample": "first %d reads of batch %d of the same workload, oracle/libsalt_oracle.so (bit-exact CPU restatement "
                                             "of the reference), %d threads = every hardware thread of this box, align time only (%.1f s)" % (ns, timed_batch, cores, cpu_dt),
                                   "speedup_1gpu": round(value / world / (ns / cpu_dt / 1e6), 1)}
            # the REAL reference binary on the same index files and a bounded sample of the same reads, when it is there (oracle/_ref/salt:
            # built from /root/reference by oracle/Makefile in the build container, shipped as a binary; test infrastructure like the
            # oracle): its own [alnse_core] clock (index reload excluded, SAM to /dev/null included, as the reference reports itself)
            ref_bin = os.path.join(ROOT, "oracle", "_ref", "salt")
            if os.access(ref_bin, os.X_OK):
                try:
                    nr = min(ns, 500000)
                    fqp = os.path.join(w["dir"], "cpu_sample.fq")
                    with open(fqp, "wb") as f:
                        f.write(workload.fastq_bytes(hs[:nr * L], nr, L, first_id=0))
                    # the reference's own thread pool gets slower with more threads on this box (16: 3.1 s, 32: 3.8 s, 64: 7.5 s, 128: 14.5 s
                    # for the same 500 000 reads): a short sweep, the best count is the baseline
                    sweep = {}
                    for thr in [t for t in (8, 16, 32) if t <= cores] or [cores]:
                        rp = subprocess.run([ref_bin, "-d", "-c", "-t", str(thr), w["prefix"], fqp], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
                        tot = [l for l in rp.stderr.decode(errors="replace").splitlines() if l.startswith("[alnse_core]: total")]
                        if rp.returncode == 0 and tot:
                            sweep[thr] = float(tot[-1].split()[2])
                    os.unlink(fqp)
                    if sweep:
                        thr = min(sweep, key=lambda t: sweep[t]); ref_s = sweep[thr]
                        port = dict(out["cpu_baseline"])
                        out["cpu_baseline"] = {"value": round(nr / ref_s / 1e6, 5), "unit": "Mreads/s", "cores": thr, "cpu_model": cpu_model, "kind": "reference",
                                               "sample": "first %d reads of batch %d of the same workload as a FASTQ file, the reference's own binary (oracle/_ref/salt -d -c -t %d on the "
                                                         "same index files, SAM to /dev/null), its [alnse_core] total (%.1f s; index reload excluded as it reports itself); the best of "
                                                         "a thread sweep: its pool gets slower beyond 16 - 32 threads on this box" % (nr, timed_batch, thr, ref_s),
                                               "thread_sweep_s": {str(t): round(v, 2) for t, v in sweep.items()},
                                               "speedup_1gpu": round(value / world / (nr / ref_s / 1e6), 1), "port": port}
                except Exception as ex:                                     # the port's number stays
                    out["cpu_baseline"]["reference_error"] = repr(ex)[:200]
            out["parity"] = {"path": "timed step", "what": "rows left by the last step of the timed region (batch %d, workspace %d) vs the oracle, every field" % (timed_batch, last % n_streams),
                             "checked_reads": int(ns), "mismatching_reads": int(len(bad)),
                             "mapped_fraction": round(float((gres["pos"] != 0xFFFFFFFF).mean()), 5)}
            ora.close()
        # ---- paired end (BASELINE configs[3]) on the same index, same workspaces ----
        if args.pe_pairs > 0:
            ora = None
            try:
                if not args.no_cpu:
                    sys.path.insert(0, os.path.join(ROOT, "oracle"))
                    import oracle_py
                    ora = oracle_py.Oracle(w["prefix"])
                out["pe"] = pe_leg(args, cfg, idx, alns, streams, genome, site, ora, oracle_py if ora else None, dev, torch, np, salt_amd, workload, log)
            except Exception as ex:                                        # a report, not a gate: the SE line must still come out
                import traceback
                log(traceback.format_exc())
                out["pe"] = {"error": repr(ex)[:400]}
            if ora:
                ora.close()
    # the index image and the workspaces go before the end-to-end legs: the `salt` processes attach their own image, and two of them do
    # not fit one GPU (a salt that finds 70 GiB free attaches a narrower k-mer table and no context table)
    for a in alns[1:]:
        a.close()
    aln.close()
    if idx is not None:
        idx.destroy()
    # N > 1: every rank lets go of its device memory, because the end-to-end legs below are ONE `salt --gpus N` process that puts its own
    # replica of the index on every GPU of the node (in-process RCCL broadcast, salt_gpu_index_replicate)
    if rank != 0:
        del d_ress, batches, timed_rows, genome, site, pos, mask
        torch.cuda.empty_cache()
    if world > 1:
        dist.barrier()
    if rank == 0:
        del d_ress, batches, timed_rows
        torch.cuda.empty_cache()
        # ---- end to end: the drop-in binary, FASTQ text -> SAM text, wall clock (SURVEY 8d's metric; never `value`) ----
        if args.e2e_reads > 0:
            try:
                out["e2e"] = e2e_leg(args, cfg, w, genome, site, workload, torch, np, log, 1 if os.environ.get("SALT_BENCH_SAME_GPU") else world)
            except Exception as ex:                                    # the leg is a report, not a gate
                out["e2e"] = {"error": str(ex)[:300]}
        if args.e2e_pairs > 0 and args.e2e_reads > 0 and world > 1 and "error" in out.get("e2e", {}):
            out["e2e_pe"] = {"error": "skipped: the single-end `salt --gpus %d` run did not finish (see e2e.error)" % world}     # not another wait of minutes on the same cause
        elif args.e2e_pairs > 0 and args.e2e_reads > 0:
            try:
                out["e2e_pe"] = e2e_pe_leg(args, w, genome, site, workload, torch, np, log, 1 if os.environ.get("SALT_BENCH_SAME_GPU") else world)
            except Exception as ex:
                out["e2e_pe"] = {"error": str(ex)[:300]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pe_leg(args, cfg, idx, alns, streams, genome, site, ora, oracle_py, dev, torch, np, salt_amd, workload, log):
    """BASELINE.json configs[3]: 2 x 150-base pairs, `-p -a 250 -b 550`, insert N(400, 50), 3 % of the fragment-end mates damaged (9 %
    substitutions + a 2-base deletion: seed-and-verify misses them, the Smith-Waterman rescue runs) and 1 % random, on the index of the
    single-end leg.  A step = salt_gpu_align_pe_resident over one resident batch: k_pack ... k_cigar on the 2n mates, k_pair, k_sw (k_swf, k_swr, k_swtb),
    k_pe_final (+ k_cigar); results stay in HBM.  Same protocol as the single-end leg: distinct resident batches rotated over the
    workspaces / streams, serialized steps for per-kernel times, the timed region between synchronisations."""
    L, n_pairs, n_batches, steps = 150, args.pe_pairs, max(1, args.pe_batches), max(1, args.pe_steps)
    n_streams = len(alns)
    opt, _ = salt_amd.AlnOpt.from_argv(["-p", "-a", "250", "-b", "550"], cfg["k"])
    t0 = time.time()
    batches = []
    for b in range(n_batches):
        seqs, offs, _, _, _ = workload.make_pairs_hash(genome, site, n_pairs, L, seed=3, batch=b, damaged=0.03, orphan=0.01)
        batches.append((seqs, offs))
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    alns[0].set_pac(idx)
    for a in alns[1:]:
        a._pac_set = True
    d_ress = [torch.zeros(2 * n_pairs * salt_amd.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in range(n_streams)]

    def step(i, k=None, o=opt):
        k = i % n_streams if k is None else k
        s, f = batches[i % n_batches]
        alns[k].align_pe_resident(o, idx, n_pairs, L, s.data_ptr(), f.data_ptr(), d_ress[k].data_ptr(), streams[k].cuda_stream)

    for i in range(2 * n_streams):                              # warm-up: scratch of every workspace (k_sw's, the mates' loci) allocated outside the clock
        step(i)
    torch.cuda.synchronize()
    for a in alns:
        a.timing(True)
    n_serial = max(3, n_batches)
    for i in range(n_serial):
        step(i, 0)
    torch.cuda.synchronize()
    skm, sc = alns[0].kernel_ms()
    skm = {k: v / max(sc, 1) for k, v in skm.items()}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record(streams[0])
    for i in range(steps):
        step(i)
        ev[i + 1].record(streams[i % n_streams])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms, nc = {}, 0
    for a in alns:
        km, c = a.kernel_ms()
        nc += c
        for k, v in km.items():
            kms[k] = kms.get(k, 0.0) + v
    kms = {k: v / max(nc, 1) for k, v in kms.items()}
    for a in alns:
        a.timing(False)
    last = steps - 1
    rows = d_ress[last % n_streams].clone()
    over = sum(a.pe_counts()[4] for a in alns)
    pc = alns[last % n_streams].pe_counts()
    value = 2.0 * n_pairs * steps / dt / 1e6
    out = {"value": round(value, 3), "unit": "M mates/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "dtype": "u32 / i16",
           "config": {"workload": "BASELINE configs[3] restated: %d pairs of 2 x %d bases per step on the index of the single-end leg, insert N(400,50), 3 %% damaged + 1 %% random "
                                  "fragment-end mates, -p -a 250 -b 550, %d distinct resident batches rotated over %d workspaces / HIP streams, inputs and results resident "
                                  "in HBM" % (n_pairs, L, n_batches, n_streams), "pairs_per_step": n_pairs, "read_len": L},
           "step_ms": spread([ev[0].elapsed_time(e) for e in ev[1:]]),
           "kernel_ms": {k: round(v, 3) for k, v in kms.items() if v}, "kernel_ms_serialized": {k: round(v, 3) for k, v in skm.items() if v},
           "per_million_mates_ms_serialized": {k: round(v * 1e6 / (2 * n_pairs), 3) for k, v in skm.items() if v},
           "rescue_requests_per_step": int(pc[0]), "rescues_beyond_capacity": int(over), "generate_s": round(t_gen, 2)}
    # ---- counters of one serialized step -> device-layout bytes of the align kernels; k_sw: its windows, reads and result rows ----
    copt, _ = salt_amd.AlnOpt.from_argv(["-p", "-a", "250", "-b", "550"], cfg["k"])
    copt.collect_counters = 0 if args.no_counters else 1
    alns[0].counters()
    step(0, 0, copt)
    torch.cuda.synchronize()
    ctr = alns[0].counters()
    qc = alns[0].queue_counts()
    ctr["heavy_reads"] = qc[0]
    out["queue_counts"] = dict(zip(("heavy_mates", "gapped_mates", "k_gap_items", "k_cigar_items"), [qc[0], qc[2], qc[5], qc[6]]))
    spr = (L - cfg["k"]) // cfg["k"] + 1
    try:
        db = device_bytes(ctr, 2 * n_pairs, L, spr)
    except KeyError:
        db = {}
    # a rescue reads its window once forward and (to the end point) once backward as 4-bit masks or 2-bit bases, the mate, and writes a 160-byte row
    db["k_sw"] = int(pc[0]) * (2 * (550 + L) // 2 + L + 160)
    # the roofline block is for the dominant MEMORY-bound kernel; k_sw (k_swf, k_swr, k_swtb, timed together) is integer DP in registers: its block
    # (`valu`) gives the share of the chip's vector issue slots it used, from the committed PMC passes
    cand = {k: v for k, v in skm.items() if k in db and v > 0 and k != "k_sw"}
    dom = max(cand, key=lambda k: cand[k]) if cand else None
    pf = os.path.join(ROOT, "profiles", "r03", "pmc_summary_pe_%s.json" % args.workload)
    prof = {}
    if os.path.exists(pf):
        try:
            prof = json.load(open(pf))
        except Exception:
            prof = {}
    if skm.get("k_sw"):
        iss = {k: prof.get("issue", {}).get(k, {}) for k in ("k_swf", "k_swr", "k_swtb")}
        out["valu"] = {"kernel": "k_sw = k_swf + k_swr + k_swtb (+ k_swf1): forward pass of request pairs in packed 16-bit lanes, reverse pass, banded traceback of the mate rescues",
                       "ms": round(skm["k_sw"], 3), "requests": int(pc[0]), "bound": "valu (integer DP in registers; no HBM or MFMA roofline applies)",
                       "valu_issue_frac": {k: v.get("valu_issue_frac") for k, v in iss.items()},
                       "valu_wave_insts": {k: v.get("valu_wave_insts") for k, v in iss.items()},
                       "ms_rocprof": {k: v.get("kernel_ms_rocprof") for k, v in iss.items()},
                       "source": "profiles/r03/pmc_summary_pe_%s.json (SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x duration x 2.4 GHz))" % args.workload}
    if dom:
        ach = db[dom] / (skm[dom] / 1e3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": dom + (" (k_heavy_pe)" if dom == "k_heavy" else ""), "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "bytes_per_launch": int(db[dom]), "avg_launch_ms": round(skm[dom], 3),
                           "all_kernels": {k: {"bytes_per_launch": int(v), "ms": round(skm.get(k, 0.0), 3),
                                               "GBps": round(v / (skm[k] / 1e3) / 1e9, 1) if skm.get(k) else None} for k, v in db.items()},
                           "timing": "HIP events on the launch stream over %d serialized steps (kernel_ms_serialized)" % n_serial,
                           "counters": {k: int(v) for k, v in ctr.items() if k.startswith("d_")},
                           "note": "the dominant memory-bound kernel of the paired-end step; k_sw has its own block (`valu`)"}
        key = "k_heavy_pe" if dom == "k_heavy" else dom
        out["roofline"]["traffic"] = prof.get("hbm_bytes_per_launch", {}).get(key)
        out["roofline"]["issue_bound"] = prof.get("issue", {}).get(key)
        rq = prof.get("per_kernel_mean", {}).get(key, {}).get("TCC_EA0_RDREQ_sum")
        if rq:
            out["roofline"]["random_requests"] = {"per_launch": int(rq), "G_per_s": round(rq / (skm[dom] / 1e3) / 1e9, 1), "ceiling_G_per_s": RANDOM_REQ_GPS,
                                                  "frac_of_ceiling": round(rq / (skm[dom] / 1e3) / 1e9 / RANDOM_REQ_GPS, 3)}
    # ---- parity + CPU baseline on the first pe_check pairs of the last timed step's batch ----
    nchk = min(args.pe_check, n_pairs)
    if nchk > 0 and ora is not None:
        s, f = batches[last % n_batches]
        hs, ho = s[:2 * nchk * L].cpu().numpy(), f[:2 * nchk + 1].cpu().numpy().view(np.uint32)
        cores, cpu_model = cpu_info()
        oo = ora.opt()
        t1 = time.perf_counter()
        want = ora.align_pe(oo, hs, ho, opt.min_tlen, opt.max_tlen, n_threads=cores)
        cdt = time.perf_counter() - t1
        got = rows.cpu().numpy().view(salt_amd.RESULT_DTYPE)[:2 * nchk]
        bad = oracle_py.compare(got, want, pe=True)
        out["parity"] = {"path": "timed step", "checked_pairs": int(nchk), "mismatching_mates": int(len(bad)),
                         "mapped_fraction": round(float((got["pos"] != 0xFFFFFFFF).mean()), 5),
                         "rescued_mates": int(((want["seq_start"] != 0) | (want["seq_end"] != L - 1)).sum())}
        out["cpu_baseline"] = {"value": round(2 * nchk / cdt / 1e6, 5), "unit": "M mates/s", "cores": cores, "cpu_model": cpu_model, "kind": "port",
                               "sample": "the same %d pairs through the oracle's alnpe_core1 restatement on %d threads (%.1f s)" % (nchk, cores, cdt),
                               "speedup_1gpu": round(value / (2 * nchk / cdt / 1e6), 1)}
    del d_ress, batches, rows
    torch.cuda.empty_cache()
    return out


def input_dir(d, need_bytes):
    """Where the end-to-end legs put their FASTQ input: memory (/dev/shm) when it has the room, so that the gigabytes of input written a
    moment before the run are not dirty pages the kernel is still writing back while `salt` writes its SAM file (the measured leg is
    FASTQ text in the page cache -> SAM text into a file either way); else next to the index."""
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize > 2 * need_bytes + (8 << 30):
            return "/dev/shm"
    except OSError:
        pass
    return d


def remove_at_exit(*paths):
    """Input files are removed when their leg ends; this covers a leg that raised in between (a file in /dev/shm is held in memory)."""
    import atexit

    def rm():
        for f in paths:
            try:
                os.unlink(f)
            except OSError:
                pass
    atexit.register(rm)


def e2e_leg(args, cfg, w, genome, site, workload, torch, np, log, n_gpus=1):
    """`salt -d -c` on a FASTQ file of the same workload: the binary's own clock (starts when the index is loaded and attached, like the
    reference's; ends with the last SAM byte written), plus the whole process."""
    L = cfg["read_len"]
    d = w["dir"]
    n = args.e2e_reads
    fq = os.path.join(input_dir(d, n * (2 * L + 32)), "salt_bench_e2e.fq")
    remove_at_exit(fq)
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
        p = subprocess.run([salt, "-d", "-c", "-t", str(threads), "--gpus", str(n_gpus), w["prefix"], fq], stdout=fo, stderr=subprocess.PIPE, timeout=300)
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
    os.unlink(sam)
    # the same command with the SAM stream thrown away: what the pipeline does when no file system is in the way
    null_s = None
    with open(os.devnull, "wb") as fo:
        p2 = subprocess.run([salt, "-d", "-c", "-t", str(threads), "--gpus", str(n_gpus), w["prefix"], fq], stdout=fo, stderr=subprocess.PIPE, timeout=300)
    for line in p2.stderr.decode(errors="replace").splitlines():
        if p2.returncode == 0 and line.startswith("[alnse_core]: total"):
            null_s = float(line.split()[2])
    os.unlink(fq)
    return {"value": round(n / align_s / 1e6, 3) if align_s else None, "unit": "Mreads/s", "reads": n, "threads": threads, "n_gpus": n_gpus,
            "value_devnull": round(n / null_s / 1e6, 3) if null_s else None,
            "what": "salt -d -c -t %d --gpus N <idx> reads.fq > out.sam (one process, reads dealt to the GPUs' workers in chunks, one ordered SAM stream): FASTQ text in, SAM text out (%.2f GB), PCIe and host I/O included; clock = the binary's "
                    "[alnse_core] total (restarted where the reference restarts its own, behind the index reload, alnse.c:1366; workspace set-up included; ends with the "
                    "last SAM byte written; index load + attach excluded, as SURVEY 8d defines it); value_devnull: the same command with stdout on /dev/null" % (threads, sam_bytes / 1e9),
            "align_wall_s": align_s, "process_wall_s": round(wall, 2), "fastq_write_s": round(t_write, 2), "input_dir": os.path.dirname(fq), "pipeline": detail}


def e2e_pe_leg(args, w, genome, site, workload, torch, np, log, n_gpus=1):
    """`salt -d -c -p -a 250 -b 550` on two FASTQ files of 2 x 150-base pairs of the same genome (BASELINE's paired-end configuration): the
    binary's own clock as in e2e_leg."""
    L, n, d = 150, args.e2e_pairs, w["dir"]
    di = input_dir(d, 2 * n * (2 * L + 32))
    fq = [os.path.join(di, "salt_bench_e2e_1.fq"), os.path.join(di, "salt_bench_e2e_2.fq")]
    remove_at_exit(*fq)
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
        p = subprocess.run([os.path.join(ROOT, "salt_amd", "bin", "salt"), "-d", "-c", "-p", "-a", "250", "-b", "550", "-t", str(threads), "--gpus", str(n_gpus), w["prefix"]] + fq,
                           stdout=fo, stderr=subprocess.PIPE, timeout=300)
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
    os.unlink(sam)
    null_s = None
    with open(os.devnull, "wb") as fo:
        p2 = subprocess.run([os.path.join(ROOT, "salt_amd", "bin", "salt"), "-d", "-c", "-p", "-a", "250", "-b", "550", "-t", str(threads), "--gpus", str(n_gpus), w["prefix"]] + fq,
                            stdout=fo, stderr=subprocess.PIPE, timeout=300)
    for line in p2.stderr.decode(errors="replace").splitlines():
        if p2.returncode == 0 and (line.startswith("[alnpe_core]: total") or line.startswith("[alnse_core]: total")):
            null_s = float(line.split()[2])
    for f in fq:
        os.unlink(f)
    return {"value": round(2 * n / align_s / 1e6, 3) if align_s else None, "unit": "M mates/s", "pairs": n, "read_len": L, "threads": threads, "n_gpus": n_gpus,
            "value_devnull": round(2 * n / null_s / 1e6, 3) if null_s else None,
            "what": "salt -d -c -p -a 250 -b 550 -t %d <idx> r1.fq r2.fq > out.sam (%.2f GB of SAM); the binary's clock as in e2e" % (threads, sam_bytes / 1e9),
            "align_wall_s": align_s, "process_wall_s": round(wall, 2), "input_dir": di, "pipeline": detail}


if __name__ == "__main__":
    main()
