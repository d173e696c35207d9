#!/usr/bin/env python3
"""bench.py -- audio packets/s of the MI355X Vorbis audio-packet synthesis path (BASELINE.json metric).

Workload at any N: BASELINE.json configs[1] per GPU -- batches of 4096 synthetic 44.1 kHz stereo long-block
(n=2048) packets, window flags (1,1), each batch one logical stream of 4096 consecutive packets (state carried
inside the launch); 8 distinct batches are rotated so the ~50 MB a launch touches is not served from the
256 MiB Infinity Cache.  A "step" = the device synthesis stage of one batch (inverse coupling, floor-1 curve,
floor x residue, IMDCT, window/overlap-add, i16 conversion), inputs (entropy-decoded records) already resident
in HBM.  Weak scaling: every rank runs the same per-GPU workload on its own streams, no collectives in the data
path (streams are independent).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PACKETS_PER_BATCH = 4096
N_BATCHES = 8
UNIQUE_PACKETS = 512
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 4000 steps = ~70 ms of GPU time: the MI355X needs tens of milliseconds of sustained load to settle its clocks
    # (200-step runs measure 20.2 us per launch, 1000-step runs 18.9, 4000-step runs 17.7; profiles/r01_steps_sweep.txt)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-generic", action="store_true", help="time the generic kernels instead of the specialised one")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-all-cores-seconds", type=float, default=4.0,
                    help="also time the CPU port on every host core at once (one worker process per core); 0 = skip")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly from Python instead of hipGraph replay")
    ap.add_argument("--format", choices=["i16", "i16_interleaved", "f32"], default="i16",
                    help="output sample format (the BASELINE metric is quoted on planar i16 = Vec<Vec<i16>>)")
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed graph replays after the W warmup steps until the device clocks have settled")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the end_to_end object (staging-ring rate incl. PCIe)")
    ap.add_argument("--e2e-batches", type=int, default=300,
                    help="batches per end-to-end run: long enough (about half a second) for a container's CPU quota to show")
    ap.add_argument("--e2e-threads", type=int, nargs="+", default=[],
                    help="host thread counts tried for the end-to-end rate (the best is reported); default: the library's "
                         "own default (CPUs this process may use: affinity and cgroup cpu.max) and 1.25 x that")
    ap.add_argument("--gate-us", type=float, default=0.0,
                    help="length of the spin kernel in front of the timed region (see the comment at ev0); 0 = none")
    ap.add_argument("--graph-segments", default="",
                    help="K <= 512: capture the timed steps as consecutive hipGraphs of these lengths plus one for the rest, e.g. "
                         "'1,3' ('' = ONE graph, the default: measured at K = 20, every extra graph launch costs more -- ~0.7 us "
                         "per step by HIP events -- than the earlier start of the first kernel saves; profiles/r04_submission.txt)")
    ap.add_argument("--no-active-wait", action="store_true",
                    help="leave the HIP runtime's completion wait on its default (interrupt after a short spin) instead of polling: "
                         "the host learns of the end of the K steps tens of microseconds later, all of it inside the wall-clock span")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the other_configs object (BASELINE configs[2], [3], [4]-stepping, blocksize_1 = 10 / 12 / 13)")
    ap.add_argument("--other-steps", type=int, default=600, help="timed launches per entry of other_configs (at least)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group also for one process (exercises the N > 1 code path on a 1-GPU box)")
    ap.add_argument("--device-of-rank", default="",
                    help="TEST ONLY: comma-separated device ordinal per local rank (e.g. 0,0: two ranks on the one GPU of a test box, "
                         "with LW_BENCH_BACKEND=gloo -- RCCL cannot form a group of two ranks on one device); default: device = LOCAL_RANK")
    ap.add_argument("--streams", type=int, default=256,
                    help="logical streams per 4096-packet batch (each contributes 4096/streams consecutive packets)")
    args = ap.parse_args()

    if not args.no_active_wait:
        # host side of the timed region: hipDeviceSynchronize polls the completion signal (up to 1 s) instead of sleeping on the
        # interrupt.  At the driver's K = 20 the region is ~0.3 ms, and the interrupt path's wake-up latency alone was ~5 % of it.
        os.environ.setdefault("ROC_ACTIVE_WAIT_TIMEOUT", "1000000")
    # the staging rings of the end_to_end leg give every slot its own HIP stream; the runtime maps streams onto 4 hardware
    # queues by default, and with the graph's queues in the same process two slots of a ring end up in ONE queue -- their
    # upload / kernels / PCM copy then run one after the other instead of side by side (9.0 instead of 11.7 M packets/s,
    # profiles/r04_e2e_ring.txt).  A process that runs rings next to other streams wants GPU_MAX_HW_QUEUES >= 8.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for N > 1"
    device = int(args.device_of_rank.split(",")[local_rank]) if args.device_of_rank else local_rank
    torch.cuda.set_device(device)
    # "nccl" IS RCCL on ROCm: the driver's 1/2/4/8-GPU runs.  LW_BENCH_BACKEND=gloo (test only) carries the same barrier and MAX
    # all-reduce over TCP, so that the N > 1 code of this file (per-rank seeds, rank 0's line, cpu_baseline: null, the MAX over
    # the ranks' spans) also runs where the ranks have to share a device
    backend = os.environ.get("LW_BENCH_BACKEND", "nccl")
    red_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1 or args.force_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # host threads for the (untimed) entropy stage that prepares the records: an equal share of the cores per rank
    from lewton_amd import _native as N
    # (the CPUs this process may actually use: affinity mask and cgroup quota -- 16 of the GPU box's 256 hardware threads)
    host_threads = max(2, N.lw_default_host_threads() // max(1, world))
    from lewton_amd import audio, header, streamgen as sg
    from lewton_amd.batch import Batch

    # ---- synthetic stream material (seeded per rank so that every GPU decodes different data)
    setup = sg.stereo_setup(44100, 8, 11)
    idp, _, stp = setup.headers()
    ident = header.read_header_ident(idp)
    st = header.read_header_setup(stp, 2, (8, 11))
    dec = audio.decoder_for(ident, st, device)
    pool = sg.make_stream(setup, "L", UNIQUE_PACKETS, seed=1000 + rank)
    rng = np.random.default_rng(77 + rank)

    stream = torch.cuda.current_stream()
    sptr = C.c_void_p(stream.cuda_stream)
    batches, outs, pwrs = [], [], []
    prime_idx, orders = [], []   # what every batch was built from (the parity check below re-decodes batch 0 on the CPU)
    S = args.streams
    assert PACKETS_PER_BATCH % S == 0
    per_stream = PACKETS_PER_BATCH // S
    for b in range(N_BATCHES):
        # prime every stream with one packet so that all 4096 packets of the batch yield samples
        spw = [audio.PreviousWindowRight() for _ in range(S)]
        prime = Batch(dec, S, args.format)
        pidx = rng.integers(0, UNIQUE_PACKETS, S)
        prime.entropy([(pool[int(i)], pw) for i, pw in zip(pidx, spw)])
        prime.upload(sptr)
        prime.synth_to_host(sptr)
        prime.close()
        bt = Batch(dec, PACKETS_PER_BATCH, args.format)
        if args.force_generic:
            bt.set_force_generic(True)
        order = rng.integers(0, UNIQUE_PACKETS, PACKETS_PER_BATCH)
        # stream-major order: stream s contributes packets [s*per_stream, (s+1)*per_stream) of the batch
        res = bt.entropy([(pool[int(i)], spw[k // per_stream]) for k, i in enumerate(order)], n_threads=host_threads)
        assert all(r[0] == 0 and r[1] == 1024 for r in res)
        pwr = spw
        bt.upload(sptr)
        out = torch.empty(bt.out_elems, dtype=torch.float32 if args.format == "f32" else torch.int16, device="cuda")
        batches.append(bt)
        outs.append(out)
        pwrs.append(pwr)
        prime_idx.append(pidx)
        orders.append(order)
    torch.cuda.synchronize()
    alg_bytes = batches[0].algorithmic_bytes  # 12 420 B per stereo long packet
    assert alg_bytes == PACKETS_PER_BATCH * (12420 + (4096 if args.format == "f32" else 0)), alg_bytes

    def step(k, sp):
        b = k % N_BATCHES
        batches[b].synth(C.c_void_p(outs[b].data_ptr()), outs[b].numel(), sp)

    for k in range(args.warmup):
        step(k, sptr)
    torch.cuda.synchronize()
    # One hipGraph holding N_BATCHES consecutive steps (launch-bound inner loop -> graph replay); the Python
    # interpreter would otherwise be the bottleneck at ~25 us per launch.
    graph = tail_graph = None
    # K <= 512: ONE graph holds all K steps (one submission for the whole timed region; --graph-segments splits it for
    # experiments).  Larger K: a graph of N_BATCHES steps replayed K // N_BATCHES times plus a tail graph (the submissions
    # after the first hide behind queued work)
    seg_graphs = []
    per_graph = args.steps if args.steps <= 512 else N_BATCHES
    n_tail = args.steps % per_graph
    if not args.no_graph:
        def capture(first, count):
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_):
                cs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                for k in range(first, first + count):
                    step(args.warmup + k, cs)
            return g_
        if args.steps <= 512:
            segs, done = [], 0
            for tok in [t for t in args.graph_segments.split(",") if t.strip()]:
                n_ = int(tok)
                if n_ > 0 and done + n_ < args.steps:
                    segs.append((done, n_))
                    done += n_
            segs.append((done, args.steps - done))
            seg_graphs = [capture(a_, n_) for a_, n_ in segs]
            graph = seg_graphs[-1]
        else:
            graph = capture(0, per_graph)
            if n_tail:  # K is not a multiple of the graph length: the remainder is its own graph, not eager launches
                tail_graph = capture(0, n_tail)
        torch.cuda.synchronize()
        # untimed: let the device clocks settle (DVFS reaches its steady state only after tens of ms of load, far
        # longer than W steps of 18 us); the timed region below is still exactly K steps
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            for _ in range(16):
                graph.replay()
            torch.cuda.synchronize()
    # spin kernel in front of the timed region (see below): calibrate torch.cuda._sleep's tick once, aim at --gate-us
    gate_ticks = 0
    if graph is not None and hasattr(torch.cuda, "_sleep") and args.gate_us > 0:
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000)
        torch.cuda.synchronize()
        c0.record(stream)
        torch.cuda._sleep(20000)
        c1.record(stream)
        torch.cuda.synchronize()
        us_per_tick = max(c0.elapsed_time(c1) * 1e3 / 20000, 1e-5)
        gate_ticks = max(1, int(args.gate_us / us_per_tick))
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    # The HIP-event span below is device time of the K steps: a short spin kernel (~17 us) goes first so that the first graph
    # is already queued when the first event is reached (otherwise the ~10 us the host needs to submit it sit between ev0
    # and the first kernel: +0.5 us per step at K = 20).  The wall clock t0..t1 includes the spin instead of that wait.
    if gate_ticks:
        torch.cuda._sleep(gate_ticks)
    ev0.record(stream)
    k = 0
    if seg_graphs:
        for g_ in seg_graphs:
            g_.replay()
        k = args.steps
    elif graph is not None:
        while k + per_graph <= args.steps:
            graph.replay()
            k += per_graph
        if tail_graph is not None:
            tail_graph.replay()
            k += n_tail
    while k < args.steps:
        step(args.warmup + k, sptr)
        k += 1
    ev1.record(stream)
    ev1.synchronize()         # (waits on the one event; the device-wide synchronize the contract asks for then has nothing left to wait for)
    torch.cuda.synchronize()
    # this rank's own span ends HERE, before the closing barrier: at K = 20 the timed region is ~0.35 ms, and an RCCL
    # barrier (tens of microseconds plus rank skew) inside it would be charged to the kernels at N > 1 only.  The job's
    # time is the MAX over the ranks' spans (shard.max_elapsed below); the barrier after it only re-aligns the ranks.
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    from lewton_amd import shard
    if args.force_dist and world == 1:  # run the collective of the N > 1 path once
        chk = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(chk, op=dist.ReduceOp.MAX)
        assert float(chk.item()) == elapsed
    own_elapsed = elapsed
    elapsed = shard.max_elapsed(elapsed, dist if dist.is_initialized() else None, red_dev)
    # device time of the K steps on the launch stream (HIP events), per step
    launch_ms = ev0.elapsed_time(ev1) / args.steps

    # ---- parity of the bytes that were just timed: outs[0] as the last timed replay of batch 0 left it, against the
    #      oracle (rank 0): streams 0, S/2 - 1 and S - 1 of that batch, every packet of each, starting from the
    #      primed state (the oracle decodes the priming packet first, exactly as the GPU stream did)
    parity = None
    kernels = batches[0].last_kernels
    if rank == 0 or world > 1:   # (N > 1: every rank checks the bytes it timed; rank 0 reports all of them, `ranks` below)
        try:
            from oracle import pyoracle as po
            o_id = po.Ident(idp)
            o_st = po.Setup(stp, o_id)
            ofmt = {"i16": "i16", "i16_interleaved": "i16_itl", "f32": "f32"}[args.format]
            host = outs[0].cpu().numpy()
            per_pkt = 2 * 1024                     # elements per packet block (every timed packet yields 1024 x 2)
            chk_streams = list(range(S))      # all of them: 4096 packets cost the oracle ~0.2 s
            ok, n_chk = True, 0
            for sidx in chk_streams:
                opw = po.Pwr()
                po.read_audio_packet(o_id, o_st, pool[int(prime_idx[0][sidx])], opw, ofmt)
                for k in range(sidx * per_stream, (sidx + 1) * per_stream):
                    w = po.read_audio_packet(o_id, o_st, pool[int(orders[0][k])], opw, ofmt)
                    g = host[k * per_pkt:(k + 1) * per_pkt]
                    ok &= bool(np.array_equal(g.reshape(-1), np.asarray(w).reshape(-1)))
                    n_chk += 1
            parity = ("timed batch 0: all %d streams x %d packets = %d packets %s bit-exact vs oracle" % (
                len(chk_streams), per_stream, n_chk, args.format)) if ok else "MISMATCH in timed batch 0"
        except Exception as e:  # the oracle is only a checker here
            parity = "unchecked: %r" % (e,)

    per_rank = None
    if world > 1:   # what each rank decoded and found: seeds, its own span, its parity string
        mine = {"rank": rank, "device": device, "pool_seed": 1000 + rank, "order_seed": 77 + rank, "elapsed_s": own_elapsed, "parity": parity}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # (the contract: rank 0 at N=1 only; null in the N>1 lines)
        from oracle import pyoracle as po
        o_id = po.Ident(idp)
        o_st = po.Setup(stp, o_id)
        reps, secs, npk = 1, 0.0, 0
        # calibrate on one pass, then run ~cpu_seconds of single-threaded work (lewton is single-threaded per stream)
        _, _, s1 = po.decode_stream_i16(o_id, o_st, pool, keep=False)
        reps = max(1, int(args.cpu_seconds / max(s1, 1e-3)))
        tt = time.perf_counter()
        _, _, secs = po.decode_stream_i16(o_id, o_st, pool * reps, keep=False)
        npk = len(pool) * reps
        cpu = {"value": npk / secs, "unit": "packets/s", "cores": 1, "kind": "port",
               "sample": "%d stereo long packets (same generator as the GPU workload), oracle/lewton_oracle.c "
                         "(C restatement of lewton incl. entropy decode), 1 thread, %.1f s" % (npk, secs)}
        try:  # per-stage split of the baseline: bit-serial entropy stage vs synthesis; informational
            e_s, t_s = po.stage_split(o_id, o_st, pool)
            cpu["entropy_share"] = round(e_s / t_s, 4)
            # like-for-like with `value` (kernel-resident = synthesis stage only): the same port without its entropy stage
            cpu["synthesis_only"] = {"value": cpu["value"] / max(1e-9, 1.0 - e_s / t_s), "unit": "packets/s", "cores": 1,
                                     "note": "cpu_baseline.value / (1 - entropy_share): stages A7-A14 alone, what the timed "
                                             "kernels replace"}
        except Exception as e:
            cpu["entropy_share"] = None
        # all host cores, one independent stream per core (separate worker processes; informational)
        if args.cpu_all_cores_seconds > 0:
            try:
                import subprocess
                import sys
                # one worker per CPU the container may use (affinity mask, cgroup cpu.max): more runnable processes than that
                # are only throttled
                ncore = max(1, N.lw_default_host_threads())
                procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_bench", str(2000 + i), str(args.cpu_all_cores_seconds)],
                                          cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for i in range(ncore)]
                tot = 0.0
                for p in procs:
                    out, _ = p.communicate(timeout=120 + 10 * args.cpu_all_cores_seconds)
                    n_, s_ = out.split()
                    tot += float(n_) / float(s_)
                cpu["all_cores"] = {"value": tot, "unit": "packets/s", "cores": ncore, "kind": "port",
                                    "sample": "%d worker processes (= CPUs usable by this container of %d hardware threads), one "
                                              "synthetic stream each, %.0f s each" % (ncore, os.cpu_count() or 0, args.cpu_all_cores_seconds)}
            except Exception as e:  # informational only
                cpu["all_cores"] = {"error": repr(e)}

    # ---- end-to-end rate through the library's staging ring (informational; never `value`): host entropy stage on this
    #      box's cores -> pinned staging -> H2D -> kernels -> D2H, entropy decode of batch N+1 overlapping the GPU work of N
    e2e_obj = None
    if rank == 0 and world == 1 and not args.no_end_to_end:
        try:
            from lewton_amd import e2e as e2e_mod
            best = None
            from lewton_amd import _native as N_
            cpus = N_.lw_default_host_threads()
            for thr in (args.e2e_threads or [cpus, cpus + cpus // 4]):
                r_ = e2e_mod.measure(dec, pool, n_batches=args.e2e_batches, packets=PACKETS_PER_BATCH, streams=S, threads=thr,
                                     slots=3, callers=1, samples=args.format)
                if best is None or r_["value"] > best["value"]:
                    best = r_
            e2e_obj = best
            e2e_obj["host_cpus_usable"] = cpus   # hardware threads cut to the affinity mask and the cgroup CPU quota
            try:   # entropy stage on the device (k_entropy, one wave per packet): the packets themselves cross PCIe, two host
                   # threads read prologues and plan; at the bench's batch size and with larger batches in flight
                keys = ("value", "unit", "records", "h2d_GBps", "d2h_GBps", "host_entropy_stage_alone", "host_threads", "ring_slots",
                        "kernels", "packets")
                r_ = e2e_mod.measure(dec, pool, n_batches=args.e2e_batches, packets=PACKETS_PER_BATCH, streams=S, threads=2, slots=3,
                                     callers=1, samples=args.format, device_entropy=True)
                e2e_obj["device_entropy"] = {k: r_[k] for k in keys}
                e2e_obj["device_entropy"]["packets_per_batch"] = PACKETS_PER_BATCH
                big = 4 * PACKETS_PER_BATCH
                r_ = e2e_mod.measure(dec, pool, n_batches=max(12, args.e2e_batches // 2), packets=big, streams=S, threads=0, slots=3,
                                     callers=1, samples=args.format, device_entropy=True)
                e2e_obj["device_entropy"]["large_batches"] = dict({k: r_[k] for k in keys}, packets_per_batch=big)
            except Exception as e:
                e2e_obj["device_entropy"] = {"error": repr(e)}
            try:   # one process, every GPU of the node (lw_sharder_*: a staging ring per device, streams sharded stream_id mod G,
                   # no collective); a one-GPU box runs two logical shards on the same device.  As many packets in all as the
                   # single ring above decodes (e2e_batches x 4096), so that fill and drain weigh the same in both
                ndev = max(1, N_.lw_device_count())
                devs = list(range(ndev)) if ndev > 1 else [0, 0]
                r_ = e2e_mod.measure_sharder(ident, st, pool, devs, n_calls=max(8, args.e2e_batches // len(devs)), packets_per_shard=PACKETS_PER_BATCH,
                                             streams_per_shard=S, threads=0, samples=args.format, device_entropy=True)
                e2e_obj["sharder"] = r_
            except Exception as e:
                e2e_obj["sharder"] = {"error": repr(e)}
        except Exception as e:
            e2e_obj = {"error": repr(e)}

    # ---- the other BASELINE configurations and block sizes under the same clock (informational; never `value`): each entry is
    #      timed like the headline (records resident in HBM, rotated batches, hipGraph replay, HIP events) and carries the
    #      oracle check of the PCM its timed launches left for batch 0
    other = None
    if rank == 0 and world == 1 and not args.no_other_configs:
        other = {}
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_configs as bc
            from lewton_amd import workloads as wl
            # (key, label, packets per launch, through the generic kernels only)
            for key, label, pk_, gen_ in (
                    ("3", "configs[2] mixed short/long", 4096, False), ("4", "configs[3] 5.1 @ 48 kHz", 4096, False),
                    ("10", "configs[4] stepping on one GPU: 10 000 streams x 4 packets", 4096, False),
                    ("12", "blocksize_1 = 10", 4096, False), ("11", "blocksize_1 = 12", 4096, False), ("13", "blocksize_1 = 13", 4096, False),
                    ("14", "mixed 512/1024 (blocksize 9 / 10, LLLSSSLLLL)", 4096, False),
                    ("15", "mixed 256/1024 (blocksize 8 / 10, LLLSSSLLLL)", 4096, False),
                    ("20", "mixed 512/4096 (blocksize 9 / 12, LLLSSSLLLL: libvorbis' sizes below ~64 kbit/s at 44.1 kHz)", 4096, False),
                    ("21", "mixed 1024/4096 (blocksize 10 / 12: no edge form, the transition blocks' overlap-add on the generic kernel)", 4096, False),
                    # round 6: SURVEY 8(d) config 3 as written (ONE stream), its all-long counterpart, a stream shape behind the
                    # canonicalising pre-pass (libvorbis' 5.1 coupling steps), the fallback every specialised kernel is measured
                    # against, and the mixed shapes at the batch size the library's own staging ring runs with
                    ("17", "configs[2] as SURVEY 8(d) words it: ONE stream x 4096 packets, LLSSSSSSSSL", 4096, False),
                    ("18", "ONE stream x 4096 long packets", 4096, False),
                    ("16", "5.1 @ 48 kHz with libvorbis' coupling steps (a channel in three steps: evaluated inside k_long's waves)", 4096, False),
                    ("19", "stereo, two long modes with their own mappings (k_prep + k_long)", 4096, False),
                    ("9", "generic fallback (stereo 8/11 long blocks, forced)", 4096, True),
                    ("3", "configs[2] mixed short/long, 16 384 packets per launch", 16384, False),
                    ("14", "mixed 512/1024, 16 384 packets per launch", 16384, False),
                    ("15", "mixed 256/1024, 16 384 packets per launch", 16384, False)):
                try:
                    w_ = wl.by_key(key, pk_ // 2 if key == "9" else pk_)
                    # nb=None: as many rotated batches as put >= 0.5 GiB of algorithmic bytes into one rotation (twice the 256 MiB
                    # Infinity Cache), at least 8 while that stays below 1.5 GiB -- the rule the headline follows
                    r_ = bc.measure(w_, steps=args.other_steps, nb=None, verify=True, distinct=16 if w_.n_streams > 1 else 1,
                                    force_generic=gen_)
                    other[label] = {"us_per_launch": r_["us_per_launch"], "packets_per_launch": r_["packets_per_launch"],
                                    "M_packets_per_s": r_["M_packets_per_s"], "algorithmic_bytes_per_launch": r_["algorithmic_bytes_per_launch"],
                                    "batches_rotated": r_["batches_rotated"], "footprint_bytes": r_["footprint_bytes"],
                                    "frac": round(r_["pct_of_8TBps"] / 100.0, 4),
                                    "state_bytes_per_launch": r_["state_bytes_per_launch"],
                                    "frac_incl_state": round(r_["pct_of_8TBps_incl_state"] / 100.0, 4),
                                    "kernels": r_["kernels"], "parity": r_["parity"],
                                    "steps": r_["steps"]}
                except Exception as e:
                    other[label] = {"error": repr(e)}
        except Exception as e:
            other = {"error": repr(e)}

    # HBM traffic of one launch from the PMC passes (tools/pmc.sh -> profiles/): measured in separate rocprofv3 runs of
    # this very command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; null if no profile is committed
    traffic, pmc_file = None, "none"
    try:
        import glob
        pmc_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))[-1]   # newest round
        pm = json.load(open(pmc_file))
        traffic = pm.get("hbm_bytes_per_launch")
    except Exception:
        pass

    if rank == 0:
        total_packets = args.steps * PACKETS_PER_BATCH * world
        value = total_packets / elapsed
        ach = alg_bytes / (launch_ms * 1e-3) / 1e9
        line = {
            "metric": "audio packets/sec (44.1 kHz stereo, 2048-pt long blocks)",
            "value": value,
            "unit": "packets/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batches of 4096 synthetic 44.1 kHz stereo long-block (n=2048) "
                                   "packets per GPU = %d streams x %d consecutive packets, 8 batches rotated, records "
                                   "resident in HBM" % (S, per_stream),
                       "streams_per_batch": S, "untimed_clock_settle_ms": 0.0 if args.no_graph else args.settle_ms,
                       "packets_per_step": PACKETS_PER_BATCH, "channels": 2, "blocksize": 2048,
                       "output": {"i16": "i16 planar", "i16_interleaved": "i16 interleaved", "f32": "f32 planar"}[args.format], "kernels": kernels, "parity": parity,
                       "parallelism": "streams sharded across GPUs, no collectives"},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE, profiles/%s)" % os.path.basename(pmc_file),
                         "kernel": kernels, "launch_ms": launch_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         # informational, not in `frac`: window state of the batch's streams crossing HBM at the launch boundary
                         "state_bytes_per_launch": batches[0].state_bytes},
            "cpu_baseline": cpu,
            "end_to_end": e2e_obj,
            "other_configs": other,
        }
        if per_rank is not None:
            line["ranks"] = per_rank
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
