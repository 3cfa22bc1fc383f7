#!/usr/bin/env python
"""bench.py -- headline benchmark of the DAS4Whales hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nx 20000] [--ns 120000]

One "step" = one pass of the hot path over one [nx x ns] float32 strain block that is already
resident in HBM: the f-k filter (dsp.fk_filter_filt) followed by the HF+LF fin-whale matched filter
(detect.compute_cross_correlogram x2, fused) -- BASELINE.json's metric "channel-samples/sec through
f-k filter + matched-filter".  --stages selects fk / mf / bp (zero-phase band-pass, run first).
Default workload = BASELINE.json configs[2] (20 000 channels x 120 000 samples, the shape the
metric's target is quoted on).

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank filters its own block
(independent channel blocks / files -- SURVEY.md 8e "replicas", BASELINE config 5), no collective
in the data path, weak scaling; the timed region is bracketed by barrier + synchronize and the
MAX over ranks is reported.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
live HIP-event timing) and `cpu_baseline` (NumPy oracle on a bounded sample, rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import contextlib
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PASS_NAMES = ["fk_passA_fwd", "fk_passC_fwd", "fk_passB_mid", "fk_passC_inv", "fk_passA_inv"]      # channel-first order
PASS_NAMES_TF = ["fk_passA_fwd", "fk_passBf_time_fwd", "fk_passCm_channel", "fk_passBi_time_inv", "fk_passA_inv"]   # time-first

# the arithmetic the path computes in: float32 everywhere; the matched filter multiplies binary16 hi / lo splits of its float32
# operands on the matrix cores and accumulates in float32 (22-23 significant bits per operand)
DTYPE = "f32 (matched filter: 2 x f16 split operands on the matrix cores, f32 accumulate)"



def _best_of(fn, n=3):
    """BASELINE.md section 3 protocol: one warm-up call, then the best of n timed calls."""
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def cpu_baseline(sample_nx, sample_ns, stages):
    """the NumPy / SciPy float64 restatement of the reference path (oracle; /root/reference does
    not exist on the GPU box) on ONE block of BASELINE configs[0] shape (4000 x 12 000) IN FULL, time.perf_counter,
    one warm-up + best of 3 per stage, design time excluded; the stage times are added like a GPU step's."""
    from oracle import d4w_oracle as orc
    fs, dx = 200.0, 2.0419046878814697
    rng = np.random.default_rng(1234)
    x = rng.standard_normal((sample_nx, sample_ns))
    total, notes = 0.0, []
    if "bp" in stages:
        dt = _best_of(lambda: orc.bp_filt(x, fs, 14, 30), 1 if sample_nx * sample_ns > 6e7 else 3)
        total += dt
        notes.append("bp_filt %.2f s" % dt)
    if "fk" in stages:
        # the mask every reference script uses (scripts/main_mfdetect.py:46-55) at the scripts' 8.17 m channel spacing
        mask = np.ascontiguousarray(orc.hybrid_ninf_filter_design((sample_nx, sample_ns), [0, 4 * sample_nx, 4], dx, fs,
                                                                  1350., 1450., 3300, 3450, 14., 30.))
        dt = _best_of(lambda: orc.fk_filter_filt(x, mask))
        total += dt
        notes.append("fk_filter_filt %.2f s" % dt)
    if "mf" in stages:
        tt = np.arange(sample_ns) / fs
        hf = orc.gen_template_fincall(tt, fs, 17.8, 28.8, 0.68)
        lf = orc.gen_template_fincall(tt, fs, 14.7, 21.8, 0.78)
        dt = _best_of(lambda: (orc.compute_cross_correlogram(x, hf), orc.compute_cross_correlogram(x, lf)))
        total += dt
        notes.append("compute_cross_correlogram x2 %.2f s" % dt)
    samples = float(sample_nx) * sample_ns
    # the matched filter in the reference's own form (detect.py:156-166: a Python row loop over
    # scipy.signal.correlate(row, full-length template, 'full', 'fft')), on a bounded row sample, next to the batched port
    ref_form = None
    if "mf" in stages:
        rows = min(sample_nx, 1000)
        dtr = _best_of(lambda: (orc.compute_cross_correlogram_reference_form(x[:rows], hf),
                                orc.compute_cross_correlogram_reference_form(x[:rows], lf)), 1)
        ref_form = {"compute_cross_correlogram_x2_s_per_block": dtr * sample_nx / rows,
                    "sample": "row loop over scipy.signal.correlate(method='fft') as detect.py:163-164, %d of %d rows timed once, "
                              "scaled to the block" % (rows, sample_nx)}
        if "fk" in stages:
            fk_s = [float(n.split()[1]) for n in notes if n.startswith("fk_filter_filt")][0]
            ref_form["value"] = samples / (fk_s + ref_form["compute_cross_correlogram_x2_s_per_block"])
            ref_form["unit"] = "channel-samples/s"
    out = {"value": samples / total, "unit": "channel-samples/s", "cores": 1, "kind": "port",
           "cores_available": len(os.sched_getaffinity(0)),
           "sample": "oracle (numpy.fft / scipy.signal float64, single thread as the reference runs) on one %d x %d block in "
                     "full, 1 warm-up + best of 3 per stage: %s" % (sample_nx, sample_ns, "; ".join(notes))}
    if ref_form is not None:
        out["reference_form"] = ref_form
    # second column (SURVEY 8d / BASELINE.md 3): the same math as a CPU would best run it -- float32, half spectrum,
    # scipy.fft on every core, one transform of the block shared by both templates
    try:
        ncores = len(os.sched_getaffinity(0))
        x32 = x.astype(np.float32)
        be, bnotes = 0.0, []
        if "fk" in stages:
            mh = orc.fold_mask_half(mask).astype(np.float32)
            dt = _best_of(lambda: orc.fk_filter_filt_best_effort(x32, mh))
            be += dt
            bnotes.append("rfft2 f-k %.3f s" % dt)
        if "mf" in stages:
            dt = _best_of(lambda: orc.compute_cross_correlogram_best_effort(x32, [hf[:136], lf[:156]]))
            be += dt
            bnotes.append("batched rfft matched filter x2 %.3f s" % dt)
        if be > 0:
            out["best_effort"] = {"value": samples / be, "unit": "channel-samples/s", "cores": ncores,
                                  "sample": "scipy.fft float32, workers = all cores, same block, 1 warm-up + best of 3: "
                                            + "; ".join(bnotes)}
    except Exception as e:                       # the second column must never break the bench line
        out["best_effort"] = {"error": repr(e)}
    return out


def replicas_step_ms(args, stages, world, rank, device, dist):
    """One independent [nx x ns] block per GPU (no collective in the data path, weak scaling): ms per step, max over ranks --
    the second line of the N > 1 bench (SURVEY 8e "replicas")."""
    import das4whales_amd as dw
    from das4whales_amd import detect as ddet
    nx, ns, fs, dx = args.nx, args.ns, 200.0, 2.0419046878814697
    gen = torch.Generator(device=device)
    gen.manual_seed(4321 + rank)
    x = torch.randn((nx, ns), dtype=torch.float32, device=device, generator=gen)
    y = torch.empty_like(x)
    plan = dw.dsp.FkPlan(nx, ns, device=device)
    mask = dw.dsp.fk_filter_design((nx, ns), [0, nx, 1], dx, fs)
    plan.set_mask(mask)
    del mask
    time_ax = np.arange(ns) / fs
    tpl = [ddet._normalised_support(ddet.gen_template_fincall(time_ax, fs, 17.8, 28.8, 0.68)),
           ddet._normalised_support(ddet.gen_template_fincall(time_ax, fs, 14.7, 21.8, 0.78))]

    def step():
        _, mean, mx = plan.apply_stats(x, out=y)
        return ddet._xcorr_device(y, tpl, normalize=True, stats=(mean, mx))
    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item()) / args.steps * 1e3


_RCCL_LOG = None


def _rccl_debug_capture(rank):
    """Before init_process_group at N > 1: have RCCL write its tuning decisions (which algorithm / protocol a collective of a given
    size runs as) to a file of this rank, unless the caller has set NCCL_DEBUG himself.  One line per collective call into a
    file: noise against an 8-GB transfer.  Read back by _rccl_all_gather_summary()."""
    global _RCCL_LOG
    if "NCCL_DEBUG" in os.environ:
        return
    _RCCL_LOG = "/tmp/d4w_rccl_rank%d_%d.log" % (rank, os.getpid())
    os.environ["NCCL_DEBUG"] = "INFO"
    os.environ["NCCL_DEBUG_SUBSYS"] = "TUNING,COLL"
    os.environ["NCCL_DEBUG_FILE"] = _RCCL_LOG


def _rccl_all_gather_summary():
    """Algorithm / protocol RCCL chose for the all-gathers of this run, from the rank's debug file: {"AllGather <bytes>": "algo RING
    proto SIMPLE", ...} for the largest few sizes; a note when this RCCL build printed nothing of the kind."""
    import re
    if not _RCCL_LOG or not os.path.exists(_RCCL_LOG):
        return "not captured (NCCL_DEBUG was set by the caller, or no debug file was written)"
    algos = {0: "TREE", 1: "RING", 2: "COLLNET_DIRECT", 3: "COLLNET_CHAIN", 4: "NVLS", 5: "NVLS_TREE"}
    protos = {0: "LL", 1: "LL128", 2: "SIMPLE"}
    found = {}
    try:
        for line in open(_RCCL_LOG, errors="replace"):
            m = re.search(r"AllGather:?\s+(\d+)\s+Bytes\s*->\s*Algo\s+(\d+)\s+proto\s+(\d+)", line)
            if m:
                nb, al, pr = int(m.group(1)), int(m.group(2)), int(m.group(3))
                found[nb] = "algo %s proto %s" % (algos.get(al, al), protos.get(pr, pr))
    except OSError as e:
        return "debug file unreadable: %r" % (e,)
    if not found:
        return "this RCCL build logged no 'AllGather: N Bytes -> Algo a proto p' lines (NCCL_DEBUG_SUBSYS=TUNING,COLL)"
    return {"AllGather %d B" % nb: found[nb] for nb in sorted(found, reverse=True)[:4]}


def bench_channel_sharded(args, stages, world, rank, device, dist):
    """BASELINE configs[3]: ONE nx x ns block sharded by contiguous channel block over the ranks; a step
    is the exact distributed f-k filter (time phase, all-to-all, channel phase, all-to-all, inverse
    time phase), the HF+LF matched filter on the local rows and, with --gather, the RCCL all-gather
    of the filtered t-x matrix.  Strong scaling: value = nx * ns / step time."""
    import das4whales_amd as dw
    from das4whales_amd import shard, detect as ddet
    nx, ns = args.nx, args.ns
    fs, dx = 200.0, 2.0419046878814697
    a, b = shard.channel_block(nx, world, rank)
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    x_loc = torch.randn((b - a, ns), dtype=torch.float32, device=device, generator=gen)
    plan = shard.ShardedFkPlan(nx, ns)
    mask = dw.dsp.fk_filter_design((nx, ns), [0, nx, 1], dx, fs)          # every rank designs the same mask
    plan.set_mask(mask)             # closed form: each rank evaluates the gains of its own sub-rows, no dense mask
    del mask
    torch.cuda.empty_cache()
    time_ax = np.arange(ns) / fs
    tpl = [ddet._normalised_support(ddet.gen_template_fincall(time_ax, fs, 17.8, 28.8, 0.68)),
           ddet._normalised_support(ddet.gen_template_fincall(time_ax, fs, 14.7, 21.8, 0.78))]

    fused_stats = plan.packed and "fk" in stages and "mf" in stages     # row mean / max|.| from the last f-k pass's epilogue

    def step(gather=None):
        gather = args.gather if gather is None else gather
        st = None
        if fused_stats:
            y, mean, mx = plan.apply(x_loc, stats=True)
            st = (mean, mx)
        else:
            y = plan.apply(x_loc) if "fk" in stages else x_loc
        # the all-gather of the filtered t-x matrix travels behind the matched filter of the local rows
        direct = args.gather_how == "direct"
        pend = shard.all_gather_rows(y, nx, async_op=True, how=args.gather_how) if (gather and (direct or nx % world == 0)) else None
        out = ddet._xcorr_device(y, tpl, normalize=True, stats=st) if "mf" in stages else None
        if pend is not None:
            pend[1].wait()
        elif gather:
            shard.all_gather_rows(y, nx)
        return out

    def timed(gather):
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(gather)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    for _ in range(args.warmup):
        step()
    dt = timed(args.gather)
    # the same step with its outputs left sharded (the t-x all-gather out of the timed region), so that lines of rounds that
    # defaulted differently stay comparable: reported beside the headline, never in its place
    dt_sharded = timed(False) if (args.gather and world > 1) else None
    nranks = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(nranks)                       # every rank that took part counts itself (RCCL)
    # per-stage times of rank 0 (HIP events on the launch stream; transfers are waited on there), a few extra steps
    stage_ms = {}
    for rep_i in range(3):
        plan.marks = []
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        st = None
        if fused_stats:
            y, mean, mx = plan.apply(x_loc, stats=True)
            st = (mean, mx)
        else:
            y = plan.apply(x_loc) if "fk" in stages else x_loc
        ev[1].record()
        if "mf" in stages:
            ddet._xcorr_device(y, tpl, normalize=True, stats=st)
        ev[2].record()
        if args.gather or world > 1:                     # both forms, timed beside the step (alone on the links, nothing overlapped)
            shard.all_gather_rows(y, nx)
        ev[3].record()
        if args.gather or world > 1:
            shard.all_gather_rows(y, nx, how="direct")
        ev[4].record()
        ev[4].synchronize()
        for k, (a_, b_) in {"fk_filter": (0, 1), "matched_filter": (1, 2), "all_gather": (2, 3), "all_gather_direct": (3, 4)}.items():
            stage_ms[k] = stage_ms.get(k, 0.0) + ev[a_].elapsed_time(ev[b_]) / 3
        if "mf" in stages and "picks_error" not in stage_ms:
            # what a deployment would gather instead of the 9.6-GB t-x matrix: the envelope picks of the local
            # correlograms, reassembled with global channel indices on every rank -- timed beside the step, not inside it
            # (the first repetition is a warm-up: the envelope of a new long-row shape compiles its kernels once).  Every rank
            # takes the same branch: a failure is agreed on over the group before anyone enters the collective.
            try:
                cg = ddet._xcorr_device(y, tpl, normalize=True, stats=st)
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
                pk = [ddet.pick_times_env(c_, 0.45 * float(c_.max())) for c_ in cg]
                e1.record()
                ok_local = 1
            except Exception as e:                                   # the main line must survive this side measurement
                ok_local, err_local = 0, repr(e)
            okt = torch.tensor([ok_local], dtype=torch.int32, device=device)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            if int(okt.item()) == 0:
                stage_ms["picks_error"] = err_local if not ok_local else "another rank failed"
            else:
                tabs = [shard.all_gather_picks(p_.packed, a) for p_ in pk]
                e2.record()
                e2.synchronize()
                if rep_i > 0:
                    stage_ms["picks_env_local"] = stage_ms.get("picks_env_local", 0.0) + e0.elapsed_time(e1) / 2
                    stage_ms["all_gather_picks"] = stage_ms.get("all_gather_picks", 0.0) + e1.elapsed_time(e2) / 2
                stage_ms["picks_gathered"] = int(sum(t_.shape[1] for t_ in tabs))
                del tabs
            cg = pk = None
        for (l0, e0), (l1, e1) in zip(plan.marks[:-1], plan.marks[1:]):
            stage_ms["fk:" + l1] = stage_ms.get("fk:" + l1, 0.0) + e0.elapsed_time(e1) / 3
    plan.marks = None
    rep_ms = None
    plan_info = {"N1": plan.N1, "N2": plan.N2, "sub_rows_owned": plan.nq, "sub_rows_per_rank": plan.sub_rows_per_rank(),
                 "channel_phase_balance": round(plan.channel_phase_balance(), 4), "packed": bool(plan.packed),
                 "exchange_row_chunks": plan.exchange_chunks() if plan.packed else None}
    gather_info = {"in_timed_step": ("t-x matrix (all_gather_rows, how=%s)" % args.gather_how) if args.gather else "none: outputs stay sharded",
                   "all_gather_tx_ms": stage_ms.get("all_gather"), "all_gather_tx_direct_ms": stage_ms.get("all_gather_direct"),
                   "forms": {"collective": "one dist.all_gather_into_tensor; RCCL's choice below (a ring all-gather is bound by ONE xGMI "
                                           "link: >= 55 ms for 8.4 GB at N = 8, SURVEY 8e)",
                             "direct": "N - 1 grouped isend / irecv pairs per rank, all point-to-point links busy (>= 7.8 ms for the same bytes)"},
                   "rccl_all_gather": _rccl_all_gather_summary() if world > 1 else "n/a at N = 1",
                   "tx_bytes_into_every_gpu": 4.0 * nx * ns * (world - 1) / world,
                   "picks_env_local_ms": stage_ms.get("picks_env_local"), "all_gather_picks_ms": stage_ms.get("all_gather_picks"),
                   "picks_gathered": stage_ms.get("picks_gathered")}
    kernels = "shape-specialised" if plan.packed else "generic"
    if (world > 1 or args.force_replicas) and stages == ["fk", "mf"] and not args.no_replicas:
        del x_loc, plan
        torch.cuda.empty_cache()
        rep_ms = replicas_step_ms(args, stages, world, rank, device, dist)
    if rank == 0:
        samples = float(nx) * ns
        ms = dt / args.steps * 1e3
        gbs = 24.0 * samples / (ms * 1e-3) / 1e9 / world           # per-GPU algorithmic f-k bytes over the whole step
        out = {"metric": "channel-samples/sec through f-k filter + matched-filter", "value": samples / (dt / args.steps),
               "unit": "channel-samples/s", "n_gpus": world, "rccl_ranks": int(nranks.item()), "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE,
               "data": "synthetic",
               "config": {"workload": "BASELINE configs[3]: ONE %d channels x %d samples float32 block sharded by channel block over %d GPU(s), "
                                      "classic f-k fan mask, stages %s%s" % (nx, ns, world, "+".join(stages),
                                                                             ", all-gather of the t-x output" if args.gather else ""),
                          "plan": plan_info,
                          "parallelism": "channel blocks x%d, pencil f-k (2 all-to-all)" % world},
               "roofline": {"bound": "hbm", "kernel": "distributed step (%s pass kernels + exchange)" % kernels,
                            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                            "traffic": None, "stage_ms_rank0": stage_ms},
               "gather": gather_info}
        out["value_note"] = ("value / ms_per_step INCLUDE the RCCL all-gather of the filtered t-x matrix (BASELINE configs[3])"
                             if args.gather else "value / ms_per_step EXCLUDE the t-x all-gather (outputs stay sharded; --gather puts it in)")
        if dt_sharded is not None:
            out["outputs_sharded"] = {"ms_per_step": dt_sharded / args.steps * 1e3, "value": samples / (dt_sharded / args.steps),
                                      "unit": "channel-samples/s", "note": "same step without the t-x all-gather"}
        if rep_ms is not None:
            out["replicas"] = {"note": "second line: one independent block per GPU, no collective in the data path (weak scaling)",
                               "ms_per_step": rep_ms, "value": samples * world / (rep_ms * 1e-3), "unit": "channel-samples/s",
                               "scaling": "weak"}
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


def bench_stream(args, world, rank, device, dist):
    """BASELINE configs[4]: consecutive 60-s files streamed through the whole detection chain -- fused ingest, band-pass
    carried across file boundaries (halo), per-file f-k filter, HF + LF matched filter whose last lags continue into the
    next file, envelope picks, spectrogram-correlation detector.  N GPUs: the record is dealt in runs of consecutive
    files (rank r owns files [r F, (r + 1) F), SURVEY 8e "consecutive files to the same GPU"), so only the run boundaries
    cross ranks: one raw halo each way (exchanged up front, the raw data being resident) and one filtered head that rank
    r + 1 sends as soon as its first file is filtered and rank r needs only for its last file.  Weak scaling."""
    import das4whales_amd as dw
    from das4whales_amd import data_handle, detect as ddet, dsp as ddsp, stream
    nx = args.nx if args.nx else 11020
    ns = args.ns if args.ns else 12000
    fs, dx, halo = 200.0, 2.0419046878814697, 1024
    F = max(2, args.files)
    sel = [0, nx, 1]
    meta = {"scale_factor": 1.7e-11 * 1e9, "fs": fs, "dx": dx}     # nano-strain: the unit factor rides on the ingest kernel's scale
    mask = ddsp.hybrid_ninf_filter_design((nx, ns), [0, nx * 4, 4], dx, fs, 1350., 1450., 3300, 3450, 14., 30.)
    t = np.arange(ns) / fs
    hf = ddet.gen_template_fincall(t, fs, 17.8, 28.8, 0.68)
    lf = ddet.gen_template_fincall(t, fs, 14.7, 21.8, 0.78)
    lmax = max(len(ddet._normalised_support(hf)), len(ddet._normalised_support(lf)))
    kernel = {"f0": 27., "f1": 17., "dur": 0.8, "bdwidth": 4.}

    tap_hf, tap_lf = ddet._normalised_support(hf), ddet._normalised_support(lf)
    ncall, span = 6, 160                               # notes injected per file, channels a note is seen on

    def raw_file(i):
        """Synthetic raw int32 file i of the record (same on whichever rank makes it): white noise (sigma 3e4 counts) plus
        `ncall` fin-whale notes (HF / LF alternating) seen on `span` adjacent channels each with the linear moveout of a
        2 km/s apparent speed at the mask's 8.17 m spacing -- inside the f-k pass band, so that the matched filter has
        something to find and "detections" are detections, not threshold crossings of noise."""
        g = torch.Generator(device=device).manual_seed(1000 + i)
        x = torch.randn((nx, ns), device=device, generator=g) * 3e4
        rs = np.random.default_rng(1000 + i)
        for k in range(ncall):
            tp = torch.from_numpy(np.ascontiguousarray(tap_hf if k % 2 == 0 else tap_lf, dtype=np.float32)).to(device)
            rc = int(rs.integers(span, max(span + 1, nx - span)))
            t0 = int(rs.integers(200, ns - 600))
            rows = torch.arange(max(0, rc - span // 2), min(nx, rc + span // 2), device=device)
            start = t0 + torch.round((rows - rc).abs().float() * (8.17 / 2000.0 * fs)).long()
            cols = start[:, None] + torch.arange(tp.numel(), device=device)[None, :]
            x[rows[:, None], cols] += 3e4 * tp[None, :]
        return x.to(torch.int32)

    def strain(raw):
        x, _, _ = data_handle.load_das_data_array(raw, sel, meta)
        return x

    first_file = rank * F
    raws = [raw_file(first_file + j) for j in range(F)]
    ingest = host_raws = None
    h2d_gbs = None
    if args.from_host:
        # the files start in PINNED HOST memory (what a reader that fills data_handle.PinnedIngest.host_array, or its own
        # pinned pool, leaves behind; the read itself is I/O and not timed): file i + 1 crosses PCIe on the side stream
        # while file i is processed
        ingest = data_handle.PinnedIngest((nx, ns), np.int32, device=device, depth=2)
        host_raws = [r.cpu().pin_memory() for r in raws]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ingest.upload(0, host_raws[0]); ingest.raw(0)
        e0.record(); ingest.upload(1, host_raws[1 % F]); ingest.raw(1); e1.record(); e1.synchronize()
        h2d_gbs = nx * ns * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    # raw halos across the run boundaries (set-up, not timed: the raw record is resident)
    prev_tail = next_head = None
    if world > 1:
        my_head = strain(raws[0])[:, :halo].contiguous()
        my_tail = strain(raws[-1])[:, -halo:].contiguous()
        prev_tail = torch.empty_like(my_tail) if rank > 0 else None
        next_head = torch.empty_like(my_head) if rank < world - 1 else None
        ops = []
        if rank < world - 1:
            ops += [dist.P2POp(dist.isend, my_tail, rank + 1), dist.P2POp(dist.irecv, next_head, rank + 1)]
        if rank > 0:
            ops += [dist.P2POp(dist.isend, my_head, rank - 1), dist.P2POp(dist.irecv, prev_tail, rank - 1)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    det_streams = [torch.cuda.Stream(device) for _ in range(max(0, min(2, int(getattr(args, "detector_streams", 0)))))]

    def run():
        pend = {}
        nxt_filt = torch.empty((nx, lmax - 1), dtype=torch.float32, device=device) if (world > 1 and rank < world - 1) else None
        if nxt_filt is not None:
            pend["recv"] = dist.irecv(nxt_filt, rank + 1)

        def on_filtered(idx, y):
            if idx == 0 and world > 1 and rank > 0:          # the previous rank's last file waits for this head
                pend["head"] = y[:, :lmax - 1].contiguous()
                pend["send"] = dist.isend(pend["head"], rank - 1)

        def filtered_head():
            pend["recv"].wait()
            return nxt_filt

        st = stream.FileStream(fs, 14, 30, templates=[hf, lf], fk_mask=mask, halo=halo, prev_tail=prev_tail, on_filtered=on_filtered)
        npicks = 0

        picks, keep = [], []
        main = torch.cuda.current_stream(device)

        def detect_on(done):
            # nothing here waits for the device: the threshold 0.45 max(corr) (scripts/main_mfdetect.py:82,95) is formed on the
            # device from the correlator's row maxima, the pickers are launched and their tables are sized when the run looks
            # at them (below, inside the timed region).  The two detectors of a file are independent of each other and of the
            # next file's filters: with --detector-streams 2 (default) the envelope picks and the spectrogram correlation
            # run on a stream each beside the filter chain (their kernels are one workgroup per row, bound by latency
            # chains, not by bytes: they fill each other's gaps); the results of the run stay referenced until the streams
            # are joined, so no block is handed out again while another stream still reads it.
            for r in done:
                keep.append(r)
                for sd in det_streams:
                    sd.wait_stream(main)
                with torch.cuda.stream(det_streams[0]) if det_streams else contextlib.nullcontext():
                    rm = r.get("row_max")      # per-row maxima from the correlator's epilogue (the last file of a run: one read)
                    thr = ddet.Threshold(0.45, ddet.correlogram_max(r["correlograms"][0], rm[0] if rm else None, on_device=True))
                    cs = r["correlograms"]
                    both = ddet.stacked_pair(cs[0], cs[1]) if len(cs) == 2 and os.environ.get("D4W_BENCH_STACK", "1") != "0" else None
                    # the HF and LF correlograms of a file sit back to back (one fused correlator launch): their envelopes and
                    # picks at the common threshold are one launch each over 2 nx rows
                    for c in ([both] if both is not None else cs):
                        picks.append(ddet.pick_times_env(c, thr, lazy=True))
                with torch.cuda.stream(det_streams[-1]) if det_streams else contextlib.nullcontext():
                    keep.append(ddet.compute_cross_correlogram_spectrocorr(r["filtered"], fs, [14., 30.], kernel, 0.8, 0.95))
            return 0
        if ingest is not None:
            ingest.upload(0, host_raws[0])
            for j in range(F):
                if j + 1 < F:
                    ingest.upload((j + 1) & 1, host_raws[j + 1])         # waits (on the device) for the slot's last consumer
                x, _, _ = ingest.strain(j & 1, sel, meta)
                npicks += detect_on(st.push(x))
        else:
            for raw in raws:
                npicks += detect_on(st.push(strain(raw)))
        npicks += detect_on(st.flush(next_head=next_head, next_filtered_head=filtered_head if nxt_filt is not None else None))
        if "send" in pend:
            pend["send"].wait()
        npicks += sum(p.total for p in picks)          # sizes and packs every pick table of the run (one 16-byte copy each)
        for sd in det_streams:
            main.wait_stream(sd)
        return npicks

    for _ in range(max(1, args.warmup // 2)):
        run()
    torch.cuda.synchronize()
    if ingest is not None:
        ingest.timing = []
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    npicks = 0
    reps = max(1, args.steps // 5)
    for _ in range(reps):
        npicks += run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt, float(npicks)], dtype=torch.float64, device=device)
        dist.all_reduce(tt[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(tt[1:], op=dist.ReduceOp.SUM)
        dt, npicks = float(tt[0]), float(tt[1])
    if rank == 0:
        nfiles = reps * F * world
        out = {"metric": "channel-samples/sec through the streaming detection chain (ingest + band-pass + f-k + matched filter + picks + spectrogram correlation)",
               "value": nfiles * float(nx) * ns / dt, "unit": "channel-samples/s", "n_gpus": world, "steps": reps * F, "warmup": args.warmup,
               "ms_per_step": dt / (reps * F) * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE,
               "data": "synthetic", "files_per_s": nfiles / dt, "detections_per_s": npicks / dt, "detections": npicks,
               "injected_notes_per_file": ncall, "channels_per_note": span, "detector_streams": len(det_streams),
               "ingest": ({"from": "pinned host memory, double-buffered upload on a side stream (data_handle.PinnedIngest)",
                           "h2d_GBps_one_file": h2d_gbs, "file_bytes": nx * ns * 4,
                           "h2d_ms_per_file_under_load": float(np.mean([a_.elapsed_time(b_) for a_, b_ in ingest.timing])) if ingest.timing else None,
                           "pcie_bound_files_per_s": h2d_gbs * 1e9 / (nx * ns * 4)} if ingest is not None
                          else {"from": "device-resident raw files"}),
               "config": {"workload": "%d consecutive 60-s files of %d channels x %d samples per GPU (int32 raw), halo %d samples, "
                                      "hybrid_ninf f-k mask, HF+LF templates, envelope picks at 0.45 max, spectrogram correlation"
                                      % (F, nx, ns, halo), "parallelism": "runs of consecutive files x%d, halo hand-off between neighbours" % world}}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def _self_launch(n, backend):
    """Re-run this command line under torch.distributed.run with n ranks on this node (127.0.0.1 rendezvous, a free
    port); returns the launcher's exit code.  Fails loudly when the node has fewer than n GPUs."""
    import socket
    import subprocess
    if backend == "nccl" and torch.cuda.device_count() < n:
        print("[bench] --gpus %d requested but %d GPU(s) visible; refusing to print a smaller line" % (n, torch.cuda.device_count()),
              file=sys.stderr, flush=True)
        sys.exit(2)
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    rc = subprocess.call(cmd, env=env)
    sys.exit(rc)


def bench_gloo_emulated(args, stages, world, rank):
    """--backend gloo: the launch path and the channel-sharded f-k step of the N > 1 bench on CPU ranks -- the same
    shard.ShardedFkPlan exchange code over gloo with the emulator build of the kernel sources (tests/emu) standing in for
    libd4w.so.  Test infrastructure for tests/test_bench_launch.py (no GPU in the build container); prints the same JSON
    line shape with "data": "synthetic (emulated, CPU)" so that it can never be mistaken for a measurement."""
    import ctypes
    import importlib.util
    import torch.distributed as dist
    from tests.emu_util import load_emu
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    emu = load_emu()
    vp_, ci = ctypes.c_void_p, ctypes.c_int
    emu.d4w_fkd_plan_create.argtypes = [ci] * 4 + [ctypes.POINTER(vp_)]
    emu.d4w_fkd_plan_destroy.argtypes = [vp_]
    emu.d4w_fkd_plan_info.argtypes = [vp_, ctypes.POINTER(ci)]
    emu.d4w_fkd_plan_q1_owner.argtypes = [vp_, ctypes.POINTER(ci)]
    emu.d4w_fkd_set_mask_dense_f32.argtypes = [vp_] * 3
    emu.d4w_fkd_time_fwd_f32.argtypes = [vp_] * 3 + [ci, vp_]
    emu.d4w_fkd_chan_apply_f32.argtypes = [vp_] * 3
    emu.d4w_fkd_time_inv_f32.argtypes = [vp_] * 3
    emu.d4w_fkd_time_fwd_packed_f32.argtypes = [vp_] * 3 + [ci, vp_]
    emu.d4w_fkd_time_inv_packed_f32.argtypes = [vp_] * 4
    emu.d4w_fkd_time_fwd_packed_rows_f32.argtypes = [vp_] * 3 + [ci, ci, ci, vp_]
    emu.d4w_fkd_time_inv_packed_rows_f32.argtypes = [vp_] * 3 + [ci, ci, vp_]

    def check(rc):
        if rc != 0:
            raise RuntimeError(emu.d4w_last_error())
    spec = importlib.util.spec_from_file_location("d4w_shard", os.path.join(ROOT, "das4whales_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    nx, ns = args.nx or 100, args.ns or 600
    a, b = shard.channel_block(nx, world, rank)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn((nx, ns), generator=g)
    m = torch.rand((nx, ns), generator=g)
    plan = shard.ShardedFkPlan(nx, ns, native=(emu, check))
    plan.set_mask(m)
    x_loc = x[a:b].contiguous()

    def step():
        return plan.apply(x_loc)
    for _ in range(args.warmup):
        step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ranks = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(ranks)
    # the two ways of collecting the result, timed beside the step as in the GPU line: the filtered t-x matrix, or picks
    # (here: the samples above 3 sigma of the local rows, as a packed 2 x K (local row, time index) table)
    t1 = time.perf_counter()
    y_all = shard.all_gather_rows(y, nx)
    t2 = time.perf_counter()
    y_dir = shard.all_gather_rows(y, nx, how="direct")           # N - 1 grouped isend / irecv pairs per rank
    t2d = time.perf_counter()
    same_forms = bool(torch.equal(y_all, y_dir))
    hit = torch.nonzero(y > 3.0 * y.std()).t().contiguous()
    t2p = time.perf_counter()
    tab = shard.all_gather_picks(hit, a)
    t3 = time.perf_counter()
    err = None
    if rank == 0 and nx * ns <= 1 << 20:
        from oracle import d4w_oracle as orc
        ref = orc.fk_filter_filt(x.double().numpy(), m.double().numpy())
        err = float(np.max(np.abs(y_all.numpy() - ref)) / np.max(np.abs(ref)))
    if rank == 0:
        dt = float(tt.item())
        print(json.dumps({"metric": "channel-samples/sec through f-k filter", "value": float(nx) * ns * args.steps / dt,
                          "unit": "channel-samples/s", "n_gpus": world, "ranks": int(ranks.item()), "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": dt / max(args.steps, 1) * 1e3, "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic (emulated, CPU)",
                          "config": {"workload": "ONE %d x %d block sharded by channel block over %d CPU rank(s), gloo, emulator "
                                                 "kernels -- launch-path test, not a measurement" % (nx, ns, world),
                                     "parallelism": "channel blocks x%d, pencil f-k (2 all-to-all)" % world,
                                     "plan": {"N1": plan.N1, "N2": plan.N2, "sub_rows_per_rank": plan.sub_rows_per_rank(),
                                              "channel_phase_balance": round(plan.channel_phase_balance(), 4)}},
                          "gather": {"in_timed_step": "none: outputs stay sharded", "all_gather_tx_ms": (t2 - t1) * 1e3,
                                     "all_gather_tx_direct_ms": (t2d - t2) * 1e3, "direct_equals_collective": same_forms,
                                     "all_gather_picks_ms": (t3 - t2p) * 1e3, "picks_gathered": int(tab.shape[1])},
                          "rel_err_vs_oracle": err}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", type=str, default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL, one rank per GPU (the bench); gloo = CPU ranks with the emulator kernels (launch-path test only)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nx", type=int, default=0, help="channels (default 20000; --config stream: 11020)")
    ap.add_argument("--ns", type=int, default=0, help="samples (default 120000; --config stream: 12000)")
    ap.add_argument("--config", type=str, default="block", choices=["block", "stream"],
                    help="block = the f-k + matched-filter step on one resident block (BASELINE configs[2] / [3]); stream = "
                         "BASELINE configs[4]: consecutive 60-s files through the whole detection chain, files/s and detections/s")
    ap.add_argument("--files", type=int, default=8, help="--config stream: consecutive files per GPU")
    ap.add_argument("--from-host", action="store_true",
                    help="--config stream: the raw files start in pinned host memory and cross PCIe double-buffered on a side "
                         "stream (data_handle.PinnedIngest) instead of being device-resident")
    ap.add_argument("--plan", type=str, default="", help="C1,C2,N1,N2,TA,TC override")
    ap.add_argument("--cpu-sample", type=str, default="4000x12000", help="CPU baseline block (BASELINE configs[0] shape, run in full)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--detector-streams", type=int, default=0,
                    help="--config stream: HIP streams the two detectors of a file run on beside the filter chain (0, the default since "
                         "round 6: one stream for everything -- 201 against 188 files/s: the band-pass kernel owns its compute units' LDS "
                         "now and shares them with nothing; 2 was the default of round 5)")
    ap.add_argument("--no-dense", action="store_true", help="skip the extra dense-mask / hybrid_ninf f-k timings")
    ap.add_argument("--prune-eps", type=float, default=4e-6,
                    help="opt-in tail pruning threshold of the fk_hybrid_ninf_pruned block (relative to the mask maximum)")
    ap.add_argument("--stages", type=str, default="fk,mf", help="comma list of bp, fk, mf")
    ap.add_argument("--api", type=str, default="public", choices=["public", "private"],
                    help="'public' (default since round 6): the timed step is the package's PUBLIC calls -- dsp.bp_filt, "
                         "dsp.fk_filter_filt(x, mask), detect.compute_cross_correlograms(y, [hf, lf]) with the full-length "
                         "zero-padded templates the reference's scripts build (the DC tail of detect.py:158 included) -- i.e. "
                         "exactly what the parity tests certify; 'private': the composition of rounds 1-5 (FkPlan.apply_stats + "
                         "detect._xcorr_device on support-only templates, no tail term); the other one is timed beside it")
    ap.add_argument("--no-fused-stats", action="store_true",
                    help="matched-filter row statistics by a separate pass over the filtered block instead of the f-k epilogue")
    ap.add_argument("--shard", type=str, default="auto", choices=["auto", "replicas", "channel"],
                    help="'channel' = ONE block sharded by channel block over the GPUs: exact distributed f-k filter (two "
                         "exchanges) + matched filter on the local rows + RCCL all-gather of the filtered t-x matrix -- "
                         "BASELINE configs[3], strong scaling, the default for N > 1; 'replicas' = one independent block "
                         "per GPU (no collective, weak scaling); auto = single-device step at N = 1, 'channel' otherwise")
    ap.add_argument("--gather", dest="gather", action="store_true", default=None,
                    help="--shard channel: all-gather the filtered t-x matrix inside every timed step (default: the filtered rows and "
                         "their correlograms stay with their rank; the t-x all-gather and the gather of the picks are timed "
                         "separately and reported under roofline.stage_ms_rank0 / gather)")
    ap.add_argument("--no-gather", dest="gather", action="store_false")
    ap.add_argument("--gather-how", type=str, default="collective", choices=["collective", "direct"],
                    help="--shard channel: how the filtered t-x matrix is reassembled inside the timed step: 'collective' = one "
                         "dist.all_gather_into_tensor (RCCL picks the algorithm; a ring is bound by ONE xGMI link, SURVEY 8e), "
                         "'direct' = N - 1 grouped isend / irecv pairs per rank, every point-to-point link busy "
                         "(shard.all_gather_rows(how='direct')); the other form is timed beside the step and both are printed")
    ap.add_argument("--no-replicas", action="store_true", help="N > 1: skip the second (replicas, weak-scaling) measurement")
    ap.add_argument("--force-replicas", action="store_true", help="run the second measurement at N = 1 as well (exercises the N > 1 code path)")
    args = ap.parse_args()
    stages = [t for t in args.stages.split(",") if t]
    assert set(stages) <= {"bp", "fk", "mf"} and stages, "--stages: comma list of bp, fk, mf"

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU under
        # torch.distributed.run) instead of printing an n_gpus = 1 line; the children's rank 0 prints the JSON line
        return _self_launch(args.gpus, args.backend)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus > 1 or world > 1:
            print("[bench] --gpus %d but the launcher started %d rank(s): reporting n_gpus = %d" % (args.gpus, world, world),
                  file=sys.stderr, flush=True)
        args.gpus = world
    if args.backend == "gloo":
        return bench_gloo_emulated(args, stages, world, rank)
    if torch.cuda.device_count() < max(1, min(world, local_rank + 1)):
        print("[bench] rank %d: local rank %d but only %d GPU(s) visible" % (rank, local_rank, torch.cuda.device_count()),
              file=sys.stderr, flush=True)
        sys.exit(2)
    if args.shard == "auto":
        args.shard = "channel" if world > 1 else "replicas"
    if args.gather is None:
        # BASELINE configs[3] / north star: "a single RCCL all-gather over xGMI to reassemble the filtered t-x matrix" is part
        # of the sharded step; the step with its outputs left sharded is timed beside it (outputs_sharded)
        args.gather = world > 1
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.shard == "channel":
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        import datetime
        if world > 1 and rank == 0:
            _rccl_debug_capture(rank)
        # a rank that dies must not leave the others waiting for the default half hour
        dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=5))
    if args.config == "stream":
        return bench_stream(args, world, rank, device, dist)
    args.nx = args.nx or 20000
    args.ns = args.ns or 120000
    if args.shard == "channel":
        try:
            return bench_channel_sharded(args, stages, world, rank, device, dist)
        except Exception as e:                           # a multi-GPU box still gets a line: one block per GPU, no collective
            if world == 1:
                raise
            print("[bench] rank %d: channel-sharded step failed (%r); falling back to --shard replicas" % (rank, e),
                  file=sys.stderr, flush=True)
            args.shard = "replicas"
            torch.cuda.empty_cache()

    import das4whales_amd as dw
    nx, ns = args.nx, args.ns
    fs, dx = 200.0, 2.0419046878814697
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    x = torch.randn((nx, ns), dtype=torch.float32, device=device, generator=gen)
    y = torch.empty_like(x)
    opts = [int(v) for v in args.plan.split(",")] if args.plan else None
    # the plan the public calls use for this shape and device (dsp.get_fk_plan); --plan builds a private one (private api only)
    if opts is not None and args.api == "public":
        raise SystemExit("--plan overrides need --api private (the public calls own their plan)")
    plan = dw.dsp.FkPlan(nx, ns, opts=opts, device=device) if opts is not None else dw.dsp.get_fk_plan(nx, ns, device)
    # the mask of BASELINE's configs: dsp.fk_filter_design defaults (speed fan 1400/1450/3400/3500
    # m/s), designed on the device by the product's own design kernel
    mask = dw.dsp.fk_filter_design((nx, ns), [0, nx, 1], dx, fs)
    plan.set_mask(mask)
    live_rows = plan.live_rows()
    main_order = plan.order()
    del mask
    torch.cuda.empty_cache()

    from das4whales_amd import detect as ddet, dsp as ddsp
    time_ax = np.arange(ns) / fs
    tpl = [ddet._normalised_support(ddet.gen_template_fincall(time_ax, fs, 17.8, 28.8, 0.68)),
           ddet._normalised_support(ddet.gen_template_fincall(time_ax, fs, 14.7, 21.8, 0.78))]
    import scipy.signal as sps
    sos_bp = sps.butter(8, [14 / (fs / 2), 30 / (fs / 2)], "bp", output="sos")

    # the full-length templates of the reference's scripts (scripts/main_mfdetect.py:64-69: zero beyond the call's duration)
    tpl_full = [ddet.gen_template_fincall(time_ax, fs, 17.8, 28.8, 0.68), ddet.gen_template_fincall(time_ax, fs, 14.7, 21.8, 0.78)]
    mask_pub = dw.dsp.fk_filter_design((nx, ns), [0, nx, 1], dx, fs)      # the same design, handed to the public call every step

    def step_private():
        cur = x
        if "bp" in stages:
            cur = ddsp._sosfiltfilt_device(cur, sos_bp, 51)
        st = None
        if "fk" in stages:
            if "mf" in stages and not args.no_fused_stats:
                # row mean / max|.| of the filtered block come out of the f-k filter's last pass
                _, mean, mx = plan.apply_stats(cur, out=y)
                st = (mean, mx)
            else:
                plan.apply(cur, out=y)
            cur = y
        if "mf" in stages:
            return ddet._xcorr_device(cur, tpl, normalize=True, stats=st)
        return cur

    def step_public():
        # the calls a notebook makes on a block that lives on the device (CUDA tensor in -> CUDA tensor out)
        cur = x
        if "bp" in stages:
            cur = dw.dsp.bp_filt(cur, fs, 14.0, 30.0)
        if "fk" in stages:
            cur = dw.dsp.fk_filter_filt(cur, mask_pub)       # (the plan recognises the mask it folded last: no re-fold)
        if "mf" in stages:
            return dw.detect.compute_cross_correlograms(cur, tpl_full)
        return cur

    step = step_public if args.api == "public" else step_private
    other = step_private if args.api == "public" else step_public

    def timed_loop(fn, n):
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t_) / n * 1e3

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    samples = float(nx) * ns
    value = samples * world / (dt / args.steps)
    # the other composition on the same box, right behind the timed region (not part of `value`)
    ms_other = None
    if opts is None:
        try:
            other()
            ms_other = timed_loop(other, max(3, args.steps // 2))
        except Exception as e:                           # noqa: BLE001
            ms_other = "failed: %r" % (e,)

    # per-stage / per-kernel timing with HIP events on the launch stream (torch's current stream
    # is the stream every d4w_* call is issued on), averaged over the same K steps
    def ev_time(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1)

    stage_ms = {}
    acc = np.zeros(5)
    mf_how = ddet._xcorr_method(tpl, ns, "auto")        # the kernel the step's matched filter runs on ("mm" unless overridden)
    mf_k = {"mf_row_stats": 0.0, "mf_xcorr_mm": 0.0, "mf_xcorr_fft_fused": 0.0, "mf_xcorr_fft_blocks_1tpl": 0.0}
    # the public step adds the zero-padded templates' DC tail inside the correlator (detect._tails_in_kernel): the kernel timed
    # below is that launch
    coefs = [ddet._tail_coef(t) for t in tpl_full]
    mf_tails = coefs if (args.api == "public" and ddet._tails_in_kernel(tpl, coefs, ns, mf_how)) else None
    for _ in range(args.steps):
        if "bp" in stages:
            stage_ms["bp_sosfiltfilt"] = stage_ms.get("bp_sosfiltfilt", 0.0) + ev_time(
                lambda: ddsp._sosfiltfilt_device(x, sos_bp, 51))
        fused = "fk" in stages and "mf" in stages and not args.no_fused_stats
        st = None
        if "fk" in stages:
            if fused:
                _, mean_f, mx_f, ms = plan.apply_stats(x, out=y, timed=True)
                st = (mean_f, mx_f)
            else:
                _, ms = plan.apply_timed(x, out=y)
            acc += np.array(ms)
        if "mf" in stages:
            src = y if "fk" in stages else x
            stage_ms["mf_xcorr" if fused else "mf_rowstats_xcorr"] = stage_ms.get(
                "mf_xcorr" if fused else "mf_rowstats_xcorr", 0.0) + ev_time(
                lambda: ddet._xcorr_device(src, tpl, normalize=True, stats=st, tails=mf_tails))
            # the stage's kernels one by one: row_stats, the two-template matrix-core kernel the step runs (ONE launch), and
            # for reference the overlap-save FFT kernels of rounds 1-3 (fused two-template launch, one-template form)
            mean = torch.empty(nx, dtype=torch.float64, device=device)
            mx = torch.empty(nx, dtype=torch.float32, device=device)
            mf_k["mf_row_stats"] += ev_time(lambda: dw._lib.check(dw._lib.lib.d4w_row_stats_f32(
                src.data_ptr(), nx, ns, mean.data_ptr(), mx.data_ptr(), torch.cuda.current_stream().cuda_stream)))
            mf_k["mf_xcorr_mm"] += ev_time(lambda: ddet._xcorr_device(src, tpl, normalize=True, method="mm", stats=(mean, mx), tails=mf_tails))
            if mf_tails is not None:                    # the same launch without the tail term (the kernel of rounds 4-5), for reference
                mf_k["mf_xcorr_mm_no_tail"] = mf_k.get("mf_xcorr_mm_no_tail", 0.0) + ev_time(
                    lambda: ddet._xcorr_device(src, tpl, normalize=True, method="mm", stats=(mean, mx)))
            mf_k["mf_xcorr_fft_fused"] += ev_time(lambda: ddet._xcorr_device(src, tpl, normalize=False, method="fft"))
            one = 0.0
            for tp in tpl:
                one += ev_time(lambda: ddet._xcorr_device(src, [tp], normalize=False, method="fft"))
            mf_k["mf_xcorr_fft_blocks_1tpl"] += one / len(tpl)
    acc /= args.steps
    stage_ms = {k: v / args.steps for k, v in stage_ms.items()}
    pass_names = PASS_NAMES_TF if main_order["order"] == "time-first" else PASS_NAMES
    kernel_ms = {pass_names[i]: float(acc[i]) for i in range(5)} if "fk" in stages else {}
    if "fk" in stages:
        stage_ms["fk_filter"] = float(acc.sum())
    # algorithmic bytes per launch: an f-k pass reads and writes the block once
    # (8 B/sample); row_stats reads it once (4 B/sample); the fused matched-filter launch reads it once
    # and writes two correlograms (12 B/sample); band-pass stage 2 x (4+4) B/sample
    alg_bytes = {n: 8.0 * samples for n in PASS_NAMES + PASS_NAMES_TF}
    cand = dict(kernel_ms)
    if "mf" in stages:
        for k in mf_k:
            kernel_ms[k] = mf_k[k] / args.steps
            if k == "mf_row_stats" and "fk" in stages and not args.no_fused_stats:
                continue                     # not part of the step: statistics come from the f-k epilogue
            if k == "mf_xcorr_mm" and mf_how != "mm":
                continue
            if k == "mf_xcorr_fft_blocks_1tpl" and (len(tpl) == 2 or mf_how != "fft"):
                continue                     # not part of the step: two templates run as ONE fused launch
            if k == "mf_xcorr_fft_fused" and (len(tpl) != 2 or mf_how != "fft"):
                continue                     # reference only unless D4W_XCORR_METHOD=fft puts it back into the step
            if k == "mf_xcorr_mm_no_tail":
                continue                     # reference only: the public step launches the kernel with the tail term
            cand[k] = kernel_ms[k]
        alg_bytes["mf_row_stats"] = 4.0 * samples
        alg_bytes["mf_xcorr_fft_blocks_1tpl"] = 8.0 * samples      # read the block, write one correlogram
        alg_bytes["mf_xcorr_fft_fused"] = 12.0 * samples           # read the block once, write two correlograms
        alg_bytes["mf_xcorr_mm"] = (4.0 + 4.0 * len(tpl)) * samples  # read the block once, write one correlogram per template
        alg_bytes["mf_xcorr_mm_no_tail"] = alg_bytes["mf_xcorr_mm"]
    if "bp" in stages:
        cand["bp_sosfiltfilt"] = stage_ms["bp_sosfiltfilt"]
        alg_bytes["bp_sosfiltfilt"] = 8.0 * samples         # read once, write once
    dom = max(cand, key=cand.get)
    achieved = alg_bytes[dom] / (cand[dom] * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")     # written by scripts/pmc_summary.py
    tkey = {"mf_row_stats": "row_stats", "mf_xcorr_fft_blocks_1tpl": "xcorr_fft_blocks",
            "mf_xcorr_fft_fused": "xcorr_fft_fused", "mf_xcorr_mm": "xcorr_mm_rows"}.get(dom, dom)
    traffic_source = None
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(tkey, {}).get("hbm_bytes_per_launch")
            # NOT counted in this run: PMC counters need their own rocprofv3 passes (one counter group per run).  The value is
            # the committed result of such a session on this kernel; the session is named so that it can be checked
            traffic_source = "profiles/pmc_traffic.json <- " + str(tj.get("_source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (scripts/pmc.sh)"))
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": alg_bytes[dom],
                "kernel_ms": kernel_ms, "stage_ms": stage_ms}
    if "fk" in stages:
        fk_gbs = 24.0 * samples / (float(acc.sum()) * 1e-3) / 1e9
        # two names for two numbers that are easy to mix up: fk_in_step_frac = the filter as the step runs it (with the row
        # statistics in the last pass's epilogue when the matched filter follows; = fk_algorithmic_frac), fk_alone_frac = the
        # same mask without that epilogue (= fk_classic.fk_algorithmic_frac, filled in below)
        roofline.update({"fk_algorithmic_GBps": fk_gbs, "fk_algorithmic_frac": fk_gbs / HBM_PEAK_GBS,
                         "fk_in_step_frac": fk_gbs / HBM_PEAK_GBS, "fk_alone_frac": None,
                         "fk_only_samples_per_s": samples / (float(acc.sum()) * 1e-3),
                         "fk_live_wavenumber_rows": live_rows, "fk_order": main_order})
        def time_mask(m, **set_kw):
            """The same filter with another mask: five pass times (HIP events) and the 24 B/sample fraction."""
            plan.set_mask(m, **set_kw)
            accm = np.zeros(5)
            plan.apply(x, out=y)
            nrep = max(2, args.steps // 2)
            for _ in range(nrep):
                _, ms = plan.apply_timed(x, out=y)
                accm += np.array(ms)
            accm /= nrep
            od = plan.order()
            names = PASS_NAMES_TF if od["order"] == "time-first" else PASS_NAMES
            return {"fk_filter_ms": float(accm.sum()), "live_wavenumber_rows": plan.live_rows(), "order": od,
                    "kernel_ms": {names[i]: float(accm[i]) for i in range(5)},
                    "fk_algorithmic_frac": 24.0 * samples / (float(accm.sum()) * 1e-3) / 1e9 / HBM_PEAK_GBS}

        if not args.no_dense:
            # the step's own mask once more WITHOUT the row-statistics epilogue the matched filter asks of the last pass
            # (fk_algorithmic_frac above includes it): the f-k filter alone, like the blocks that follow
            roofline["fk_classic"] = time_mask(dw.dsp.fk_filter_design((nx, ns), [0, nx, 1], dx, fs))
            roofline["fk_alone_frac"] = roofline["fk_classic"]["fk_algorithmic_frac"]
            # a fully dense mask (nothing skipped) and the design every reference script uses,
            # hybrid_ninf_filter_design(1350, 1450, 3300, 3450, 14, 30) (scripts/main_mfdetect.py:46-47), exact and
            # with the opt-in tail pruning (rows whose folded gain stays below prune_eps * max are treated as dead)
            dm = torch.rand((nx, ns), dtype=torch.float32, device=device, generator=gen)
            roofline["fk_dense_mask"] = time_mask(dm)
            del dm
            hm = dw.dsp.hybrid_ninf_filter_design((nx, ns), [0, nx, 1], dx, fs, 1350., 1450., 3300, 3450, 14., 30.)
            roofline["fk_hybrid_ninf"] = time_mask(hm)
            roofline["fk_hybrid_ninf_pruned"] = dict(time_mask(hm, prune_eps=args.prune_eps), prune_eps=args.prune_eps)
            del hm
            # the geometry the reference scripts run: every 4th channel of the cable (8.17 m spacing,
            # scripts/main_mfdetect.py:25-30, DAS4Whales_ExampleNotebook.md:256) -- |k| only reaches 0.061 1/m, so the
            # classic fan keeps every wavenumber row alive
            sel4 = [0, 4 * nx, 4]
            roofline["fk_classic_step4"] = dict(time_mask(dw.dsp.fk_filter_design((nx, ns), sel4, dx, fs)), selected_channels=sel4)
            roofline["fk_hybrid_ninf_step4"] = dict(time_mask(
                dw.dsp.hybrid_ninf_filter_design((nx, ns), sel4, dx, fs, 1350., 1450., 3300, 3450, 14., 30.)), selected_channels=sel4)
        if "bp" not in stages and not args.no_dense:
            # the zero-phase band-pass (dsp.bp_filt, 14-30 Hz) on the same block, outside the timed step: 8 B/sample
            # algorithmic (read once, write once)
            ddsp._sosfiltfilt_device(x, sos_bp, 51)
            tb = sum(ev_time(lambda: ddsp._sosfiltfilt_device(x, sos_bp, 51)) for _ in range(3)) / 3
            roofline["bp"] = {"bp_filt_ms": tb, "algorithmic_bytes": 8.0 * samples,
                              "frac": 8.0 * samples / (tb * 1e-3) / 1e9 / HBM_PEAK_GBS}

    if rank == 0:
        out = {"metric": "channel-samples/sec through " + " + ".join(
                   {"bp": "band-pass", "fk": "f-k filter", "mf": "matched-filter"}[t] for t in stages), "value": value,
               "unit": "channel-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
               "config": {"workload": "%d channels x %d samples float32 per GPU, classic f-k fan mask "
                                      "(fk_filter_design defaults), stages %s, HF+LF fin-call templates"
                                      % (nx, ns, "+".join(stages)),
                          "api": args.api,
                          "api_note": ("the timed step is the public calls dsp.fk_filter_filt(x, mask) + detect.compute_cross_correlograms(y, "
                                       "[hf, lf]) on full-length zero-padded templates (DC tail of detect.py:158 included)"
                                       if args.api == "public" else
                                       "the timed step is FkPlan.apply_stats + detect._xcorr_device on support-only templates (rounds 1-5)"),
                          ("ms_per_step_private_composition" if args.api == "public" else "ms_per_step_public_calls"): ms_other,
                          "plan": plan.info(), "parallelism": "independent channel blocks x%d" % world},
               "roofline": roofline}
        if world == 1 and not args.no_cpu:
            snx, sns = [int(v) for v in args.cpu_sample.split("x")]
            out["cpu_baseline"] = cpu_baseline(snx, sns, stages)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
