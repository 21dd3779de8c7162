#!/usr/bin/env python
"""Headline benchmark: meshes/sec of the GEOMetrics per-step hot path on MI355X.

One STEP = one pass of the hot path over the rank's shard of synthetic meshes
(BASELINE.json config 5 sharded: 8 meshes per GPU, 2562 verts / 5120 faces, 3000 sampled
vs 3000 GT points, 963-dim features):
    3-layer 0N-GCN (963-192-192-192, fused CSR aggregation) -> vertex positions
    -> area-weighted face sampling (fresh random draws every step)
    -> Chamfer NN (both directions) + Chamfer loss
    -> point-to-triangle scan + point-to-surface loss
    -> backward to the GCN parameters and input features
    -> (N>1: one RCCL all-reduce of the flat gradient bucket + the 8-byte loss vector)
    -> Adam step.
All inputs are resident in HBM before the timed region.  Weak scaling: per-GPU work fixed.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--meshes-per-gpu 8] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from geometrics_amd import dist as gdist  # noqa: E402
from geometrics_amd import gemm_tuning, layers, meshgen, ops, optim, utils  # noqa: E402
from geometrics_amd.chamfer_distance import chamfer_nn  # noqa: E402
from geometrics_amd.tri_distance import kd_order, tri_distance_indexed  # noqa: E402

V_LEVEL, S_PTS, G_PTS, FEAT, HID = 4, 3000, 3000, 963, 192
# The reference draws a FRESH 3000-point ground-truth subset per item per fetch (utils.py:180-182), so anything derived from the
# ground-truth cloud belongs inside the step.  The culled Chamfer tiles need an index of the cloud (ops.GtIndex: a k-d visiting
# order + run spheres, built outside the step): with that build counted the route loses (one index launch 4.3-4.9 us against the
# 3.6 us the culled tiles save), so the HEADLINE runs the brute-force tiles on a ground-truth tensor that may change every replay.
# --gt-index: the culled route, for jobs whose ground-truth clouds are static (same results bit for bit).
CULLED_CHAMFER = False
# the driver step's three surface losses on a second stream, each beside the next deformation block (driver_step_times)
DRIVER_STEP_OVERLAP = False
# ... the three stages' surface losses as one call on the 48 stacked meshes with per-mesh weights
DRIVER_STEP_STACKED_LOSSES = True
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32 vector peak == f32 MFMA dense peak

# Roofline constants (MI355X_MICROARCH.md).  The two arg-min scans are fp32-VALU work written WITHOUT fused multiply-adds
# (the pinned reference arithmetic): every executed lane-op carries one flop, so their issue ceiling is half of the
# 157.3 TFLOP/s FMA peak: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 T lane-ops/s.  The guide's own measurement of
# un-packed v_fma_f32 is 103 TFLOP/s = 51.5 T lane-ops/s: what the chip sustains in practice.
VALU_ISSUE_TERA_LANE_OPS = FP32_PEAK_TFLOPS / 2.0
VALU_MEASURED_TERA_LANE_OPS = 103.0 / 2.0
TRI_ALGO_FLOP_PER_PAIR = 60.0    # hoisted op count of the reference's decision tree per (point, triangle)
NN_FLOP_PER_PAIR = 8.0           # 3 sub, 3 mul, 2 add
PROFILE_TAG = "r06"           # the committed profiles these figures are read from / compared with
PMC_FILE = os.path.join(ROOT, "profiles", PROFILE_TAG + "_pmc_counters.json")


DP_SEQUENCE_DEFAULT = "captured"


class Workload:
    def __init__(self, dev, first_mesh, batch, seed=3041, force_dp=False, activation=F.relu, lr=1e-4, dp_sequence=None):
        # N > 1 only: "two_graphs" (graph A, the collective issued by the host between the replays, graph B beside it) or
        # "captured" (ONE graph per step with the collective inside it, on RCCL's stream forked off the capture: one replay per
        # step, the two cross-stream dependencies become graph edges) -- GEOM_DP_SEQUENCE; measured in profiles/r05_dp_fixed_cost.txt
        self.dp_sequence = dp_sequence or os.environ.get("GEOM_DP_SEQUENCE", DP_SEQUENCE_DEFAULT)
        if self.dp_sequence not in ("two_graphs", "captured"):
            raise ValueError("dp_sequence must be 'two_graphs' or 'captured'")
        if torch.distributed.is_initialized() and torch.distributed.get_backend() != "nccl":
            self.dp_sequence = "two_graphs"     # only RCCL's collectives can be captured (the gloo test hook blocks the host)
        self.act = activation          # the reference's F.relu (GEOMetrics.py); F.elu only in the smooth-activation parity test
        V, Fc = meshgen.icosphere(V_LEVEL)
        self.batch, self.nv, self.nf = batch, V.shape[0], Fc.shape[0]
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.faces = to(Fc)
        self.base = to(meshgen.jittered_batch(V, batch, first=first_mesh))
        self.gt = to(meshgen.gt_cloud(batch, G_PTS, first=first_mesh))
        # --gt-index only: static per ground-truth cloud -- a k-d visiting order and the index of the culled Chamfer scan.
        # Results do not depend on it (tests: bit-identical with and without).  Default: none, the step reads self.gt as is,
        # and a replay after self.gt.copy_(other clouds) is a step on those clouds (tests/test_bench_step_gpu.py).
        self.gt_index = None
        if CULLED_CHAMFER:
            self.gt_index = ops.GtIndex(self.gt, torch.stack([kd_order(self.gt[i]) for i in range(batch)]))
        # the faces' visiting order (k-d leaves of the face centroids, cached per face list) from the UNDEFORMED template, which
        # every rank holds: on the culled route the samples are generated in that order, so ranks that derived it from the
        # first mesh of their own shard would draw other (equally valid) samples than one process holding the whole batch
        from geometrics_amd.tri_distance import face_order
        face_order(to(V).unsqueeze(0), self.faces)
        self.info = utils.adj_init(self.faces)
        # per-mesh seeds (global mesh index): a shard holds exactly the rows the whole-batch job would hold
        self.feat = torch.stack([torch.randn(self.nv, FEAT, generator=torch.Generator(device="cpu").manual_seed(seed + first_mesh + i))
                                 for i in range(batch)]).to(dev).requires_grad_(True)
        torch.manual_seed(seed)                       # identical (replicated) parameters on every rank
        self.stack = torch.nn.ModuleList(
            [layers.Batch_Image_ZERON_GCNGCN(i, o) for i, o in ((FEAT, HID), (HID, HID), (HID, HID))]).to(dev)
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        # force_dp: take the N > 1 sequence in a 1-rank group -- the only way to run the RCCL collective inside the step on
        # a single-GPU box
        self.dp = self.world > 1 or force_dp
        # flat DP bucket: all gradients + the shard's loss sum -> exactly one all-reduce per step.  The gradients are WRITTEN
        # into it by the end-of-pass reduction launch (bind=True) and the loss by the loss reduction (loss_out): no pack launch
        self.bucket = gdist.GradBucket(self.stack.parameters(), extra=1, force_collective=force_dp, bind=True) if self.dp else None
        self.total_meshes = batch * self.world      # equal shards (bench.py: --meshes-per-gpu on every rank)
        self.seed_grad = torch.ones((), device=dev)
        self.rng = ops.manual_seed(seed, dev, mesh_offset=first_mesh)   # sampler keyed on the GLOBAL mesh index: N shards draw what one process would
        # GEOMetrics.py:73 (Adam, lr 1e-4): every parameter tensor in one launch, step count on the device
        self.opt = optim.FusedAdam(self.stack.parameters(), lr=lr)
        self.loss = None
        self.graphs = None
        # N > 1 choreography (LAB_NOTES.md section 8): the all-reduce is issued right behind the end-of-pass reduction launch and
        # travels on RCCL's stream while the first layer's input-gradient product (postponed behind the reduction:
        # layers.late_input_gradients) runs on the launch stream; the Adam step on the reduced bucket opens the NEXT step
        # (`pending`), inside its graph: no eager launch between two replays
        self.pending = False          # an all-reduced bucket is waiting for its Adam step
        self.work = None              # the collective in flight
        self.packed_late = False      # pack() had to launch copies behind the reduction launch (a gradient that did not land in its view)
        self.reached_cut = False
        self.tail_jobs = None         # the postponed products of the pass (eager step: launched by exchange())
        self._splitting = None        # (graph A, graph B) while capture() records the step: the pass is cut behind the reduction launch

    def positions(self):
        # the three layers as a stack: the boundaries between the 192-wide layers are single launches where fused.plan says so
        # (csrc/zn_stack.hip: aggregation + the next layer's product); base + 0.01 * h[..., :3] comes out of the last layer's
        # aggregation launches (forward: positions from its epilogue; backward: [0.01 * grad_pos | 0] synthesised, never
        # written or read).  (layers.weight_gradient_batching() is for deep stacks -- the deformation block's twelve equal
        # layers; for the two equal layers here it was measured at +3 us per step.)
        return layers.zero_n_stack_positions(self.feat, self.info["adj"], list(self.stack), self.act, self.base, 0.01)

    # N = 1: one step = forward_backward(step_in_backward=True): ONE graph, Adam inside the end-of-pass reduction launch.
    # N > 1: one step = [Adam on the bucket all-reduced by the PREVIOUS step] -> forward -> backward (weight-gradient
    #        partials, reduction launch -> bucket, event, first layer's input gradient) as ONE graph; the all-reduce is issued
    #        behind the event from a side stream.  finish() applies the last pending Adam step.
    def forward_backward(self, step_in_backward=False):
        import contextlib
        if self.dp and self.pending:
            self.update()
            self.pending = False
        self.opt.zero_grad()
        self.feat.grad = None
        # nothing reads a parameter gradient before backward() returns (no hooks, no DDP: the gradients land in the bucket),
        # so the bias / weight gradients of the pass are finished by ONE launch at its end -- which, in a single-process
        # step, applies Adam to them as well (step_in_backward: no optimiser launch of its own)
        self.reached_cut = False
        capturing = self._splitting is not None
        self.tail_jobs = [] if self.dp else None           # the postponed products, collected by the end-of-pass callback
        late = (layers.late_input_gradients(self._parameter_gradients_ready, collect=self.tail_jobs) if self.dp
                else contextlib.nullcontext())
        with layers.deferred_parameter_gradients(), late, (self.opt.in_backward() if step_in_backward else contextlib.nullcontext()):
            pos = self.positions()
            self.loss = utils.batch_point_to_surface(pos, self.info, self.gt, num=S_PTS, gt_index=self.gt_index,
                                                     loss_out=self.bucket.extra if self.dp else None)
            self.loss.backward(self.seed_grad)              # explicit seed: no ones_like fill launch
        if self.dp:
            if capturing:
                # graph B: launched HERE, from the thread that also launches them in an eager step (exchange()) -- the
                # callback runs on autograd's thread, whose library handle would be created inside the capture
                for job in self.tail_jobs:
                    job()
                self.tail_jobs = None
            # normally pack() launches nothing: everything was written in place by the reduction launch
            self.packed_late = self.bucket.pack() or not self.reached_cut

    def _parameter_gradients_ready(self):
        """Called inside the backward pass (end-of-pass callback), right behind the reduction launch that wrote the bucket
        and in front of the postponed input-gradient product.  While capture() records the step: END graph A here and BEGIN
        graph B -- the collective is issued between their replays.  (ONE graph with an event-record node at this place
        would be the natural form; measured and rejected in round 4, LAB_NOTES.md section 8; round 5 captures the collective itself instead.)"""
        if self._splitting is not None:
            ga, gb = self._splitting
            ga.capture_end()
            gb.capture_begin(pool=ga.pool(), capture_error_mode="relaxed")
            self._splitting = None
        self.reached_cut = True

    def exchange(self, tail=None):
        """ONE all-reduce per step: 259 200 gradients + the shard's loss (1.04 MB).  It is started behind what the launch
        stream holds so far -- the reduction launch that wrote the bucket -- and runs on RCCL's stream while `tail` (graph B,
        or the eager step's postponed products) runs on the launch stream; the launch stream then waits for the
        collective, in front of the next step's Adam.  Two cross-stream dependencies per step, the same as a collective
        that overlaps with nothing."""
        if not self.dp:
            return
        if tail is None and self.tail_jobs:
            tail = _EagerTail(self.tail_jobs)
        if tail is not None and self.packed_late:           # a late copy into the bucket sits behind the cut: no overlap
            tail.replay()
            tail = None
        work = self.bucket.all_reduce_async()
        if tail is not None:
            tail.replay()
        if work is not None:
            work.wait()
        self.pending = True

    def update(self):
        if self.dp:
            self.opt.step(self.bucket.views, grad_scale=1.0 / self.world)   # mean over ranks folded into Adam
        else:
            self.opt.step()

    def step(self):
        self.forward_backward(step_in_backward=not self.dp)
        self.exchange()
        if not self.dp:
            self.update()      # (a no-op: the reduction launch of the pass applied it)

    def finish(self):
        """Apply the Adam step a data-parallel run still owes (the update of step s opens step s + 1)."""
        if self.dp and self.pending:
            self.update()
            self.pending = False

    def mean_loss(self):
        if self.dp:
            return float(self.bucket.extra[0]) * self.batch / self.total_meshes      # sum of the shards' means, equal shards
        return float(self.loss.detach())

    def capture(self, warm=3):
        """Record the step into HIP graphs so that no python runs between its launches (library GEMMs, our C-ABI kernels,
        Adam).  N = 1: ONE graph, the whole step.  N > 1: graph A = [Adam of the previous step, forward, backward, reduction
        launch -> bucket], graph B = [the first layer's input gradient]; the collective itself stays OUTSIDE the captured
        region (nothing depends on RCCL's graph-capture support): it is issued between the two replays from a side stream and
        runs beside graph B."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warm):
                self.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if not self.dp:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):      # the stream the warm-up steps ran on (their AccumulateGrad nodes live there)
                self.forward_backward(step_in_backward=True)
                self.update()
            self.graphs = (g,)
            return
        if not self.pending:
            raise RuntimeError("capture() of a data-parallel step needs at least one warm-up step: the captured step opens "
                               "with the Adam update of the step before it")
        if self.dp_sequence == "captured":
            # ONE graph: [Adam of the previous step, forward, backward, reduction launch -> bucket, all-reduce on RCCL's stream
            # (forked off the capture behind the reduction launch), the postponed product beside it, join]
            g = torch.cuda.CUDAGraph()
            cap = torch.cuda.Stream()
            cap.wait_stream(torch.cuda.current_stream())
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            try:
                with torch.cuda.stream(cap):
                    g.capture_begin(capture_error_mode="relaxed")
                    try:
                        self.forward_backward()
                        self.exchange()
                    finally:
                        g.capture_end()
                torch.cuda.current_stream().wait_stream(cap)
                self.pending = True       # capturing executed nothing: the update at the head of the graph is still owed
                self.graphs = (g,)
                return
            except Exception as exc:      # a stack whose collective cannot be captured: the two-graph sequence, said loudly
                print("bench.py: the collective could not be captured into the step graph (%s: %s); two graphs per step with "
                      "the all-reduce between them" % (type(exc).__name__, str(exc)[:200]), file=sys.stderr)
                from geometrics_amd import _lib
                _lib.clear_hip_error()
                torch.cuda.synchronize()
                self.dp_sequence, self.pending, self.work = "two_graphs", True, None
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        with torch.cuda.stream(cap):
            # "relaxed": the end-of-pass callback may run on autograd's device thread, and a capture begun in any other mode
            # must be ended by the thread that began it
            ga.capture_begin(capture_error_mode="relaxed")
            self._splitting = (ga, gb)
            try:
                self.forward_backward()
            except BaseException:
                (ga if self._splitting is not None else gb).capture_end()
                self._splitting = None
                raise
            if self._splitting is not None:      # the pass never reached the callback: nothing to overlap with
                ga.capture_end()
                self._splitting, gb = None, None
            else:
                gb.capture_end()
        torch.cuda.current_stream().wait_stream(cap)
        self.pending = True               # capturing executed nothing: the update recorded at the head of graph A is still owed
        self.graphs = (ga, gb)

    def run(self):
        if self.graphs is None:
            self.step()
        else:
            self.graphs[0].replay()
            if self.dp and len(self.graphs) == 2:
                self.exchange(self.graphs[1])


class _EagerTail:
    """The postponed products of a step as eager launches (same interface as a graph: replay())."""

    def __init__(self, jobs):
        self.jobs = jobs

    def replay(self):
        for job in self.jobs:
            job()


def settle_clocks(dev, ms):
    """Device conditioning in front of the W warm-up steps: `ms` milliseconds of a neutral library GEMM (NOT the workload).
    The chip leaves its idle power state over the first ~100 ms of load -- the setup in front of the timed region (mesh
    generation, eager capture steps) leaves it idle -- and a K = 20, W = 5 run (15 ms of GPU time) is otherwise timed on
    the ramp: measured on one box, ms per step at K = 20 / W = 5: 0.4935 without, 0.4749 after 20 ms, 0.4561 after 100 ms,
    0.4559 after 250 ms; steady state (K = 200, W = 20) 0.4550.  0 switches it off."""
    if ms <= 0:
        return
    a = torch.randn(4096, 4096, device=dev)
    t_end = time.perf_counter() + ms * 1e-3
    while time.perf_counter() < t_end:
        for _ in range(8):
            torch.mm(a, a)
        torch.cuda.synchronize()


def time_steps(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    gdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    gdist.barrier()
    return time.perf_counter() - t0


def event_time_us(fn, iters=30, warm=5):
    """Average duration of fn's launches: `iters` launches are captured into one HIP graph and the
    replay is bracketed by HIP events on the launch (= torch current) stream, so host launch overhead
    is excluded and what is measured is kernel time."""
    # warm-up and capture on ONE side stream: autograd creates a leaf's AccumulateGrad node on the stream of its first backward
    # and keeps it while a graph is alive; warming up on the default stream and capturing on torch's internal one left every
    # captured backward with a stream-mismatch warning (and a cross-stream wait inside the graph)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    # thread_local: in a multi-rank run RCCL's watchdog thread polls its events while this thread captures; under the default
    # ("global") mode that poll invalidates the capture
    with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
        for _ in range(iters):
            fn()
    for _ in range(3):      # a few milliseconds of the same load first: the engine clock is still ramping up behind a sync
        graph.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    graph.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


_pmc_cache = None


def pmc_table():
    """Per-kernel rocprofv3 counters committed under profiles/ (tools/pmc_traffic.sh), or {} when the file is missing or
    was collected for OTHER kernel sources: it is stamped with the digest of geometrics_amd/csrc + include, and a
    mismatch means the kernels changed since -- stale counters are never reported."""
    global _pmc_cache
    if _pmc_cache is None:
        _pmc_cache = {}
        try:
            with open(PMC_FILE) as f:
                table = json.load(f)
            from geometrics_amd import build as hip_build
            if table.get("_meta", {}).get("source_sha256") == hip_build.source_digest():
                _pmc_cache = table
        except (OSError, ValueError):
            pass
    return _pmc_cache


def pmc_row(kernel_prefix):
    for name, row in pmc_table().items():
        if name.startswith(kernel_prefix):
            return row
    return None


FETCH_16B_FACTOR = 2.0     # MI355X_MICROARCH.md, HBM: on gfx950 rocprofv3's FETCH_SIZE reports exactly 1/2 of the bytes of a
                           # wide coalesced streaming read (16 B / lane): doubled before it is compared with a byte count


def pmc_traffic_bytes(kernel_prefix, fetch_factor=FETCH_16B_FACTOR):
    """HBM-side bytes per launch = fetch_factor x FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, separate passes; KiB per launch
    in the committed file), or None without valid counters.  fetch_factor = 2 for kernels whose reads are 16 B per lane
    (the guide's gfx950 correction -- consistent with the aggregation kernel, whose 10.5 MB compulsory pass-through read
    alone exceeds the raw 8.4 MB the counter shows); WRITE_SIZE is taken as reported."""
    row = pmc_row(kernel_prefix)
    if row is None or "FETCH_SIZE_KB_per_launch" not in row:
        return None
    return int((fetch_factor * row["FETCH_SIZE_KB_per_launch"] + row["WRITE_SIZE_KB_per_launch"]) * 1024)


def valu_rate(kernel_prefix, launch_us):
    """Executed fp32-VALU work of a kernel: SQ_INSTS_VALU (wave instructions, PMC) x 64 lanes / launch time."""
    row = pmc_row(kernel_prefix)
    if row is None or "SQ_INSTS_VALU_per_launch" not in row:
        return None
    lane_ops = row["SQ_INSTS_VALU_per_launch"] * 64
    rate = lane_ops / (launch_us * 1e-6) / 1e12
    return {"valu_instructions_per_launch": int(row["SQ_INSTS_VALU_per_launch"]), "lane_ops_per_launch": int(lane_ops),
            "tera_lane_ops_per_s": round(rate, 2), "frac_of_spec_issue_rate": round(rate / VALU_ISSUE_TERA_LANE_OPS, 4),
            "frac_of_measured_vfma_rate": round(rate / VALU_MEASURED_TERA_LANE_OPS, 4)}


def fused_scan_call(w, pos, pred, flags=0, share=None, prep=None, tail=False):
    """geom_surface_scan_f32 on the step's own tensors (outputs allocated once): prep launch + the fused NN / tri launch.
    share = an earlier call object whose buffers (incl. the prepared tri workspace) are reused.  prep = what the step's draw
    launch returned (ops.draw_samples(..., prepare_scan_for, gt_index)): samples in visiting order + their index + the
    triangle records -- the call then runs the step's own variant, the culled Chamfer tiles, on `pred` = those samples."""
    import ctypes
    from geometrics_amd import _lib as L
    from geometrics_amd.tri_distance import face_order
    lib = L.lib()
    b, n_gt, num, nv, nf, dev = w.batch, G_PTS, S_PTS, w.nv, w.nf, pos.device
    f32, i32 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev)
    ws_bytes = lib.geom_tri_distance_workspace_bytes(b, n_gt, nf)
    if share is not None:
        o, ws, order, u, v, tri_order = share.keep[:6]
    else:
        o = [torch.empty(b, n_gt, **f32), torch.empty(b, n_gt, **i32), torch.empty(b, num, **f32), torch.empty(b, num, **i32),
             torch.empty(b, n_gt, **f32), torch.empty(b, n_gt, **i32), torch.empty(b, n_gt, **i32), torch.empty(b, n_gt, **f32),
             torch.empty(b, n_gt, 3, **f32), torch.empty(b, n_gt, 3, **f32)]
        ws = torch.empty(ws_bytes // 4, **f32)
        order = torch.empty(lib.geom_surface_order_words(b, nf, num, n_gt), **i32)
        u, v = torch.rand(b, num, device=dev), torch.rand(b, num, device=dev)
        tri_order = face_order(pos, w.faces)
    cull = None
    if prep is not None:      # the step's variant: culled Chamfer tiles on the samples the draw launch generated in visiting order
        choices, u, v, pred, scan_prep = prep
        ws, flags = scan_prep.tri_ws, flags | L.FLAG_TRI_WS_READY
        cull = L.SurfaceCull(w.gt_index.order.data_ptr(), w.gt_index.index.data_ptr(), scan_prep.sample_index.data_ptr(), None)
    wrote = ctypes.c_int(0)
    # tail: the finalize pass inside the launch (what the step's own call does: ops.scan_finalize_tail); needs the draws' faces
    loss = torch.empty((), dtype=torch.float32, device=pos.device)
    tail_arg = L.SurfaceTail(prep[0].data_ptr(), 1.0, 1.0, 1, loss.data_ptr(), 0) if tail and prep is not None else None

    def call():
        L.check(lib.geom_surface_scan_f32(b, n_gt, w.gt.data_ptr(), num, pred.data_ptr(), o[0].data_ptr(), o[1].data_ptr(),
                                          o[2].data_ptr(), o[3].data_ptr(), nv, pos.data_ptr(), nf, w.faces.data_ptr(),
                                          tri_order.data_ptr(), o[4].data_ptr(), o[5].data_ptr(), o[6].data_ptr(),
                                          o[7].data_ptr(), o[8].data_ptr(), o[9].data_ptr(), u.data_ptr(), v.data_ptr(), 1.0, 1.0,
                                          order.data_ptr(), flags, ws.data_ptr(), ws_bytes, ctypes.byref(wrote),
                                          ctypes.byref(cull) if cull is not None else None,
                                          ctypes.byref(tail_arg) if tail_arg is not None else None, L.stream_ptr()), "geom_surface_scan_f32")
    call.keep = (o, ws, order, u, v, tri_order, cull, prep, loss, tail_arg)
    return call


def in_step_launch_us(w, steps=24):
    """The first layer's weight-gradient launch timed WHERE IT RUNS: eager steps of the workload with HIP events around
    that launch on its stream (geometrics_amd.dense.launch_probe).  The launch in front of it (the 64 us library product of
    the input gradient) keeps the queue ahead of the events, so the bracket holds the kernel and nothing else."""
    from geometrics_amd import dense
    # Eager python launches are slower than these kernels: with an empty queue every launch starts "cold" and the bracket would
    # hold its dispatch latency (~5 us, round 3's figure).  So the stream is PLUGGED first -- ~15 ms of a neutral library
    # product -- and the eager steps are issued behind the plug: the GPU then works through a backlog, launch after launch
    # back to back as in the replayed step, and the bracket holds the kernel.
    plug = torch.randn(8192, 8192, device=w.feat.device)
    for _ in range(2):
        torch.mm(plug, plug)
    dense.launch_probe = rec = []
    try:
        for _ in range(steps):
            # rank-local on purpose: only rank 0 runs these probes while the others wait at the closing barrier, so the
            # all-reduce of a data-parallel step must not be issued here (it would pair with their barrier)
            w.forward_backward(step_in_backward=not w.dp)
            w.update()
        torch.cuda.synchronize()
    finally:
        dense.launch_probe = None
    us = sorted(s.elapsed_time(e) * 1e3 for (_, cin, _, s, e) in rec if cin == FEAT)
    us = us[len(us) // 8: len(us) - len(us) // 8] or us          # trimmed mean: the first eager steps include lazy setup
    return sum(us) / len(us) if us else None


def kernel_rooflines(w):
    """Launch-level timing of the hot kernels on the step's own tensors (HIP-graph replay bracketed by HIP events on the
    launch stream), combined with the committed PMC counters when they belong to these kernel sources."""
    from geometrics_amd import _lib
    with torch.no_grad():
        pos = w.positions().contiguous()
        pred = utils.batch_sample(pos, w.faces, num=S_PTS)
        scan = fused_scan_call(w, pos, pred)
        t_scan_all = event_time_us(scan)                                              # prep + fused launch, brute-force Chamfer tiles
        scan()                                                                        # workspace now holds this mesh's records
        t_scan_plain = event_time_us(fused_scan_call(w, pos, pred, _lib.FLAG_TRI_WS_READY, share=scan))   # the fused launch alone
        t_scan = t_scan_step = t_scan_plain
        if w.gt_index is not None:      # the step's own variant: culled Chamfer tiles on samples generated in visiting order
            prep = ops.draw_samples(pos, w.faces, S_PTS, with_points=True, prepare_scan_for=G_PTS, gt_index=w.gt_index)
            if isinstance(prep[4], ops.ScanPrep) and prep[4].sample_index is not None:
                t_scan = event_time_us(fused_scan_call(w, pos, pred, share=scan, prep=prep))
                # ... and with the loss's finalize pass riding in the launch (ops.scan_finalize_tail: what the step runs)
                t_scan_step = event_time_us(fused_scan_call(w, pos, pred, share=scan, prep=prep, tail=True))
        t_prep_tri = event_time_us(lambda: tri_distance_indexed(w.gt, pos, w.faces))  # prep + tri-only scan
        t_tri_flat = event_time_us(lambda: tri_distance_indexed(w.gt, pos, w.faces, order=None))
        from geometrics_amd.tri_distance import tri_distance as tri_soup
        corners = [pos[:, w.faces[:, i]].contiguous() for i in range(3)]      # what utils.py:467-469 hands to tri_dist
        t_tri_soup = event_time_us(lambda: tri_soup(w.gt, *corners))
        t_nn = event_time_us(lambda: chamfer_nn(w.gt, pred, 0))
        t_nn_fma = event_time_us(lambda: chamfer_nn(w.gt, pred, _lib.FLAG_NN_FMA))
        t_scan_fma = event_time_us(fused_scan_call(w, pos, pred, _lib.FLAG_NN_FMA))
        sup = torch.randn(w.batch, w.nv, HID, device=pos.device)
        csr = layers.adjacency_csr(w.info["adj"])
        t_agg = event_time_us(lambda: layers._ZeroNAggregate.apply(sup, w.stack[1].bias, csr, HID // 3, 1))
    b = w.batch
    tri_pairs = b * G_PTS * w.nf
    nn_pairs = 2 * b * G_PTS * S_PTS
    prep_row = pmc_row("tri_prep_grouped_kernel")
    algo_flops = tri_pairs * TRI_ALGO_FLOP_PER_PAIR + nn_pairs * NN_FLOP_PER_PAIR
    executed = valu_rate("surface_scan_kernel", t_scan)
    scan_bytes = b * (G_PTS * 12 + w.nv * 12 + w.nf * 24 + G_PTS * 12) + b * (G_PTS + S_PTS) * (12 + 8)
    scan_traffic = pmc_traffic_bytes("surface_scan_kernel")
    scan = {
        "kernel": "surface_scan_kernel: Chamfer NN tiles (both directions) + point-to-triangle tiles (two-level culled scan) in "
                  "one heterogeneous launch",
        "bound": "valu",
        "pipe": "fp32 VALU issue, un-fused arithmetic (arg-min scans whose per-pair roundings are pinned, not contractions)",
        "valu_issue_utilisation": None if executed is None else {
            "achieved_tera_lane_ops_per_s": executed["tera_lane_ops_per_s"], "peak_tera_lane_ops_per_s": VALU_ISSUE_TERA_LANE_OPS,
            "frac": executed["frac_of_spec_issue_rate"], "frac_of_measured_vfma_rate": executed["frac_of_measured_vfma_rate"],
            "note": "SQ_INSTS_VALU x 64 lanes / launch time: EVERY VALU instruction (cull tests, selects, address math) at "
                    "64 lanes -- an issue-slot occupancy, not useful flops"},
        "flop_roofline": {"achieved": round(algo_flops / (t_scan * 1e-6) / 1e12, 1), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(algo_flops / (t_scan * 1e-6) / 1e12 / FP32_PEAK_TFLOPS, 4),
                          "note": "SURVEY 8(d) ALGORITHMIC flops (every (point, triangle) pair at 60 + every (point, point) pair at "
                                  "8) / launch time against the 157.3 TFLOP/s fp32 peak; above 1 is possible and means "
                                  "culling, not utilisation: the tri tiles prove ~95 % of their pairs irrelevant (bit-exact "
                                  "vs brute force)"},
        "launch_us": round(t_scan, 1), "launch_us_brute_force_chamfer_tiles": round(t_scan_plain, 1),
        "launch_us_with_the_finalize_roles": round(t_scan_step, 1),
        "finalize_roles": "in the step the launch also carries the loss's finalize pass (b + 1 extra workgroups that order each mesh's "
                          "points by face as soon as ITS triangle tiles are through and sum the loss behind the last tile; "
                          "surface_finalize_kernel took 11.4 us as a launch of its own) -- the committed trace and the PMC row are "
                          "of that launch; launch_us is the scans alone",
        "chamfer_tiles": "culled (nn_culled_body: run spheres over samples the draw launch generates in face-visiting order and "
                         "over the static gt index; bit-identical to the brute-force tiles)" if t_scan is not t_scan_plain
                         else "brute force",
        "call_us_with_prep_launch": round(t_scan_all, 1), "executed": executed,
        "algorithmic_bytes_per_launch": scan_bytes,
        "traffic": scan_traffic,
        "traffic_over_algorithmic": None if scan_traffic is None else round(scan_traffic / scan_bytes, 2),
        "traffic_note": "2 x FETCH_SIZE + WRITE_SIZE; the excess over the algorithmic bytes is the triangle-record workspace "
                        "the prep launch writes and all eight XCDs re-read through their own L2 (harmless at this launch time: "
                        "< 0.2 TB/s)",
        "separate_launches_us": {"chamfer_nn": round(t_nn, 1), "tri_prep_plus_scan": round(t_prep_tri, 1),
                                 "tri_flat_scan": round(t_tri_flat, 1),
                                 "tri_reference_shaped_module (corner tensors, cached Morton order)": round(t_tri_soup, 1)},
        "fma_arithmetic_us": {"chamfer_nn (GEOM_FLAG_NN_FMA)": round(t_nn_fma, 1), "fused call": round(t_scan_fma, 1)}}
    if prep_row is not None:
        scan["prep_launch"] = {"kernel": "tri_prep_grouped_kernel", "valu_instructions_per_launch": int(prep_row.get("SQ_INSTS_VALU_per_launch", 0))}
    nn_exec = valu_rate("chamfer_nn_scalar_kernel", t_nn)
    nn_tflops = nn_pairs * NN_FLOP_PER_PAIR / (t_nn * 1e-6) / 1e12
    agg_bytes = 2 * b * w.nv * HID * 4 + csr.nnz * 12 + (w.nv + 1) * 4
    agg_gbs = agg_bytes / (t_agg * 1e-6) / 1e9
    agg_step_us = step_profile_us("zn_aggregate_ell_kernel<1, false")
    # ---- the layers' dense gradients on the fp32 matrix cores (csrc/dense_gemm.hip): the longest hand-written launches
    from geometrics_amd import dense
    rows = b * w.nv
    x1 = w.feat.detach().reshape(rows, FEAT)
    xh = torch.randn(rows, HID, device=pos.device)
    g = torch.randn(rows, HID, device=pos.device)
    w1 = w.stack[0].weight1.detach().reshape(FEAT, HID)
    wh = w.stack[1].weight1.detach().reshape(HID, HID)
    ws1 = dense.weight_workspace(rows, FEAT, HID, pos.device)
    wsh = dense.weight_workspace(rows, HID, HID, pos.device)
    dxh = torch.empty(rows, HID, device=pos.device)
    t_dw1 = event_time_us(lambda: dense.backward_weight_partials(x1, g, ws1))
    t_pair = event_time_us(lambda: dense.backward_pair(xh, g, wh, dxh, wsh))
    t_dw1_lib = event_time_us(lambda: torch.mm(x1.t(), g))
    gw1, gwh = torch.empty(FEAT, HID, device=pos.device), torch.empty(HID, HID, device=pos.device)
    t_red = event_time_us(lambda: dense.reduce([(rows, FEAT, HID, ws1, gw1, None), (rows, HID, HID, wsh, gwh, None),
                                                (rows, HID, HID, wsh, gwh, None)]))
    red_bytes = (ws1.numel() + 2 * wsh.numel()) * 4.0       # the partial tiles the launch reads (outputs: 1 MB)
    red_step_us = step_profile_us("dense_reduce_kernel")
    dw1_flop = 2.0 * rows * FEAT * HID
    pair_flop = 4.0 * rows * HID * HID
    dw1_bytes = rows * FEAT * 4 + rows * HID * 4 + ws1.numel() * 4      # X once + G once + the partial tiles written
    dw1_traffic = pmc_traffic_bytes("dense_split_kernel", fetch_factor=1.0)
    dw1_step = step_profile_us("dense_split_kernel")
    t_dw1_live = in_step_launch_us(w) or t_dw1
    roofline = {
        "kernel": "dense_split_kernel: split-K partial sums of the first layer's weight gradient dW = X^T . G "
                  "([%d, 963]^T x [%d, 192]) on v_mfma_f32_16x16x4_f32 -- the longest hand-written launch of the step" % (rows, rows),
        "bound": "mfma",
        "achieved": round(dw1_flop / (t_dw1_live * 1e-6) / 1e12, 1), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(dw1_flop / (t_dw1_live * 1e-6) / 1e12 / FP32_PEAK_TFLOPS, 4),
        "basis": "algorithmic = executed: a dense contraction, 2 * rows * 963 * 192 flop per launch, exact fp32 (no reduced "
                 "precision, nothing skipped); launch time measured live IN THE STEP: HIP events on the launch stream around "
                 "this launch in 24 eager steps of the workload issued behind a 15 ms plug (the GPU runs them back to back, "
                 "the bracket holds the kernel, not its dispatch; trimmed mean) -- the figure the committed rocprofv3 trace of "
                 "the step shows too (launch_us_in_step_profile); back to back in a graph of 30 launches the same kernel "
                 "takes launch_us_back_to_back (X from HBM every time)",
        "launch_us": round(t_dw1_live, 1), "launch_us_back_to_back": round(t_dw1, 1),
        "frac_back_to_back": round(dw1_flop / (t_dw1 * 1e-6) / 1e12 / FP32_PEAK_TFLOPS, 4),
        "launch_us_in_step_profile": dw1_step,
        "library_same_product_us": round(t_dw1_lib, 1),
        "algorithmic_bytes_per_launch": int(dw1_bytes),
        "traffic": dw1_traffic,
        "traffic_note": "FETCH_SIZE + WRITE_SIZE as reported (X is read with 4-byte loads -- its 963-float rows are never 16-byte "
                        "aligned -- so the guide's x2 for 16 B/lane reads does not apply to the bulk of the fetch)",
        "pipe_rate_note": "the matrix pipe itself sustains 155 TFLOP/s (an MFMA-only loop holds 2.36 GHz for seconds: "
                          "profiles/r04_mfma_pipe_rate.txt -- the guide's figure; round 3's '136 at 2.08 GHz' was wrong), and this "
                          "kernel runs at the full 2.39 GHz (tools/probe/kernel_clock.py).  Of its launch time 50.0 us are MFMA issue "
                          "(26 stages x 144 MFMAs x 32 cycles), ~9.5 us are what no loop tuning touches (launch, the first stages' "
                          "loads from HBM, the 18.5 MB burst of partial tiles at the end: the SAME launch with nothing but MFMAs in "
                          "its loop takes 59.5 us) and ~6 us loop inefficiency (X panel loads + LDS stores): "
                          "profiles/r04_split_kernel_ablation.txt"}
    others = {
        "surface_scan_kernel (both arg-min scans of the surface loss)": scan,
        "dense_bwd_pair_kernel (hidden layer: dX and the dW partials in ONE launch, two workgroups per CU)": {
            "bound": "mfma", "achieved": round(pair_flop / (t_pair * 1e-6) / 1e12, 1), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(pair_flop / (t_pair * 1e-6) / 1e12 / FP32_PEAK_TFLOPS, 4), "launch_us": round(t_pair, 1),
            "launch_us_in_step_profile": step_profile_us("dense_bwd_pair_kernel"),
            "library_three_launches_us_in_round2_step": 40.6},
        "dense_reduce_kernel (all weight gradients of a pass, fixed order)": {
            "bound": "hbm", "launch_us": round(t_red, 1),
            "achieved": round(red_bytes / (t_red * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(red_bytes / (t_red * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_launch": int(red_bytes),
            "in_step": None if red_step_us is None else {
                "launch_us": red_step_us, "achieved": round(red_bytes / (red_step_us * 1e-6) / 1e9, 1),
                "frac": round(red_bytes / (red_step_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "from the committed rocprofv3 trace of the step (profiles/%s_step_kernel_stats.csv): there the launch also "
                        "applies Adam (+7 MB of parameters and moments, not counted) and its partial tiles come from HBM" % PROFILE_TAG},
            "note": "`launch_us` is timed back to back on the same workspaces: the partial tiles are served from the 256 MiB "
                    "infinity cache; the in-step figure is the honest one"},
        "chamfer_nn_scalar_kernel (stand-alone launch)": {
            "bound": "valu", "pipe": "fp32 VALU, un-fused (brute force: algorithmic == executed pairs)",
            "achieved": round(nn_tflops, 2), "peak": VALU_ISSUE_TERA_LANE_OPS, "unit": "T lane-op/s (= TFLOP/s: one flop per lane-op)",
            "frac": round(nn_tflops / VALU_ISSUE_TERA_LANE_OPS, 4), "launch_us": round(t_nn, 1), "executed": nn_exec,
            "traffic": pmc_traffic_bytes("chamfer_nn_scalar_kernel")},
        "zn_aggregate_ell_kernel (forward, sign mask)": {
            "bound": "hbm", "achieved": round(agg_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(agg_gbs / HBM_PEAK_GBS, 4), "launch_us": round(t_agg, 1), "algorithmic_bytes_per_launch": agg_bytes,
            "traffic": pmc_traffic_bytes("zn_aggregate_ell_kernel<1, false"),
            "in_step": None if agg_step_us is None else {
                "launch_us": agg_step_us, "achieved": round(agg_bytes / (agg_step_us * 1e-6) / 1e9, 1),
                "frac": round(agg_bytes / (agg_step_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "from the committed rocprofv3 trace of the step (profiles/%s_step_kernel_stats.csv): operands come "
                        "from HBM there" % PROFILE_TAG},
            "note": "`launch_us` is timed back to back on one buffer pair: reads are served from the 256 MiB infinity cache; "
                    "the in-step figure is the honest one"},
    }
    return roofline, others


_step_profile = None


def step_profile_us(kernel_prefix):
    """Average duration (us) of a kernel in the committed rocprofv3 kernel trace of the step
    (profiles/<PROFILE_TAG>_step_kernel_stats.csv), or None when the file is missing / does not list it."""
    global _step_profile
    if _step_profile is None:
        _step_profile = {}
        try:
            import csv
            with open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_step_kernel_stats.csv")) as f:
                for r in csv.DictReader(f):
                    name = r.get("Name") or r.get("Kernel_Name") or ""
                    avg = r.get("AverageNs") or r.get("Average") or ""
                    if name and avg:
                        _step_profile[name.strip('"')] = float(avg) / 1e3
        except (OSError, ValueError):
            pass
    for name, us in _step_profile.items():
        if name.startswith(kernel_prefix):
            return round(us, 1)
    return None


def component_times(w):
    """Kernel time of each stage of the step on the step's own tensors (HIP-graph replay + HIP events,
    microseconds per shard of `batch` meshes).  Each stage is timed in isolation (its own launches back to back) and some
    entries nest (stack forward is part of forward+backward), so they do not add up to ms_per_step; this is the
    per-component view SURVEY 8(d) asks for."""
    from geometrics_amd import ops
    out = {}
    pos = w.positions().detach().contiguous()
    out["sampling (draws + gather/barycentric)"] = event_time_us(lambda: utils.batch_sample(pos, w.faces, num=S_PTS))
    pred = utils.batch_sample(pos, w.faces, num=S_PTS)
    out["chamfer NN, both directions"] = event_time_us(lambda: chamfer_nn(w.gt, pred))
    out["tri_distance (prep + culled scan)"] = event_time_us(lambda: tri_distance_indexed(w.gt, pos, w.faces))
    pv = pos.clone().requires_grad_(True)

    def loss_fb():
        pv.grad = None
        utils.batch_point_to_surface(pv, w.info, w.gt, num=S_PTS, gt_index=w.gt_index).backward()
    out["surface loss fwd+bwd (sampling, tri scan, NN, sums, scatter)"] = event_time_us(loss_fb, iters=10)

    def gcn_fwd():
        with torch.no_grad():
            w.positions()
    out["0N-GCN stack forward (3 GEMM + 3 aggregation)"] = event_time_us(gcn_fwd, iters=10)

    def gcn_fb():
        w.opt.zero_grad()
        w.feat.grad = None
        with layers.deferred_parameter_gradients():
            w.positions().sum().backward()
    out["0N-GCN stack forward+backward"] = event_time_us(gcn_fb, iters=10)
    for p in w.stack.parameters():
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    out["Adam (all parameter tensors, one launch)"] = event_time_us(lambda: w.opt.step(), iters=10)
    w.opt.zero_grad()
    return {k: round(v, 1) for k, v in out.items()}


# algorithmic work per mesh at BASELINE size (V = 2562, F = 5120, S = G = 3000, fp32, each operand once)
NN_PAIRS_PER_MESH = 2 * S_PTS * G_PTS                 # both directions: 18.0 M (point, point) pairs
TRI_PAIRS_PER_MESH = G_PTS * 5120                     # 15.36 M (point, triangle) pairs
LOSS_BYTES_PER_MESH = 1.4e6                           # sample 0.42 + NN 0.12 + Chamfer loss 0.20 + tri 0.16 + p2tri loss 0.47 MB
CHAMFER_BYTES_PER_MESH = 0.42e6 + 0.117e6 + 0.41e6    # sample + NN + the two-direction Chamfer loss (config 2)
GCN_CUSTOM_BYTES_PER_MESH = 24.5e6                    # 3 layers x 4.09 MB x (fwd + bwd): aggregation + epilogue traffic
GCN_GEMM_FLOP_PER_MESH = 3 * 2.0 * 2562 * (FEAT * HID + 2 * HID * HID)   # X.W, G.W^T, X^T.G of the three layers: 3.98 GFLOP


def step_level_roofline(meshes_per_s):
    """SURVEY 8(d)'s step-level figures for the full step (config 5): algorithmic bytes (loss path + the custom 0N-GCN kernels;
    the dense products' operand traffic is not in it, by the survey's definition) and algorithmic lane-ops of the two arg-min
    scans per mesh, times the measured meshes per second, against 8 TB/s and 78.6 T lane-op/s.  The products: executed
    flops against the fp32 MFMA peak."""
    return {"achieved_hbm": round((LOSS_BYTES_PER_MESH + GCN_CUSTOM_BYTES_PER_MESH) * meshes_per_s / (HBM_PEAK_GBS * 1e9), 4),
            "achieved_valu": round((NN_PAIRS_PER_MESH * NN_FLOP_PER_PAIR + TRI_PAIRS_PER_MESH * TRI_ALGO_FLOP_PER_PAIR) * meshes_per_s
                                   / (VALU_ISSUE_TERA_LANE_OPS * 1e12), 4),
            "achieved_mfma": round(GCN_GEMM_FLOP_PER_MESH * meshes_per_s / (FP32_PEAK_TFLOPS * 1e12), 4),
            "formulas": "SURVEY 8(d): hbm = (1.4 + 24.5 MB) x meshes/s / 8 TB/s; valu = (18.0 M x 8 + 15.36 M x 60 lane-ops) x meshes/s "
                        "/ 78.6 T; mfma = 3.98 GFLOP x meshes/s / 157.3 T (the scans are VALU-bound and the products MFMA-bound: "
                        "no single roof bounds the step)"}


def baseline_configs(dev):
    """BASELINE.json configs 2, 3 and 4 as such ("on 1 MI355X"), at B = 1 and at the 8-mesh shard of config 5; HIP-graph replay
    bracketed by HIP events, inputs resident, fresh draws every pass.
      2  2562-vertex icosphere, 3000 sampled vs 3000 gt points, Chamfer forward + backward (utils.batch_point_to_point)
      3  the same + tri_distance (3000 points vs 5120 faces) fused with Chamfer (utils.batch_point_to_surface), forward + backward
      4  3-layer 0N-GCN stack 963-192-192-192 on the 2562-vertex adjacency, forward + backward (to features and parameters)
    Per entry: microseconds per pass, meshes/s, the dominant kernel's roof and the pass's fraction of it by SURVEY 8(d)'s
    algorithmic counts (pair evaluations for the scans, executed flops for the products)."""
    V, Fc = meshgen.icosphere(V_LEVEL)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    faces = to(Fc)
    info = utils.adj_init(faces)
    out = {}
    for b in (1, 8):
        base = to(meshgen.jittered_batch(V, b, first=300)).requires_grad_(True)
        gt = to(meshgen.gt_cloud(b, G_PTS, first=300))

        def chamfer_fb():
            base.grad = None
            utils.batch_point_to_point(base, info, gt, num=S_PTS).backward()

        def surface_fb():
            base.grad = None
            utils.batch_point_to_surface(base, info, gt, num=S_PTS).backward()
        torch.manual_seed(3041)
        stack = torch.nn.ModuleList([layers.Batch_Image_ZERON_GCNGCN(i, o) for i, o in ((FEAT, HID), (HID, HID), (HID, HID))]).to(dev)
        feat = torch.randn(b, V.shape[0], FEAT, device=dev, requires_grad=True)

        def stack_fb():
            feat.grad = None
            for p_ in stack.parameters():
                p_.grad = None
            with layers.deferred_parameter_gradients():
                layers.zero_n_stack(feat, info["adj"], list(stack), F.relu).sum().backward()
        t2 = event_time_us(chamfer_fb, iters=10, warm=3)
        t3 = event_time_us(surface_fb, iters=10, warm=3)
        t4 = event_time_us(stack_fb, iters=10, warm=3)
        per_s = lambda t: b / (t * 1e-6)
        valu = lambda ops_per_mesh, t: round(ops_per_mesh * per_s(t) / (VALU_ISSUE_TERA_LANE_OPS * 1e12), 4)
        out["B=%d" % b] = {
            "config2_chamfer_fwd_bwd": {"us": round(t2, 1), "meshes_per_s": round(per_s(t2), 1), "dominant_kernel": "Chamfer NN tiles "
                                        "(surface_scan_kernel, two-sided)", "bound": "valu", "frac": valu(NN_PAIRS_PER_MESH * NN_FLOP_PER_PAIR, t2),
                                        "achieved_hbm": round(CHAMFER_BYTES_PER_MESH * per_s(t2) / (HBM_PEAK_GBS * 1e9), 5)},
            "config3_chamfer_tri_fused_fwd_bwd": {"us": round(t3, 1), "meshes_per_s": round(per_s(t3), 1),
                                                  "dominant_kernel": "surface_scan_kernel (point-to-triangle + Chamfer tiles)", "bound": "valu",
                                                  "frac": valu(NN_PAIRS_PER_MESH * NN_FLOP_PER_PAIR + TRI_PAIRS_PER_MESH * TRI_ALGO_FLOP_PER_PAIR, t3),
                                                  "achieved_hbm": round(LOSS_BYTES_PER_MESH * per_s(t3) / (HBM_PEAK_GBS * 1e9), 5)},
            "config4_zero_n_stack_fwd_bwd": {"us": round(t4, 1), "meshes_per_s": round(per_s(t4), 1),
                                             "dominant_kernel": "the 963-wide products (library forward / input gradient, dense_split_kernel)",
                                             "bound": "mfma", "frac": round(GCN_GEMM_FLOP_PER_MESH * per_s(t4) / (FP32_PEAK_TFLOPS * 1e12), 4),
                                             "achieved_hbm": round(GCN_CUSTOM_BYTES_PER_MESH * per_s(t4) / (HBM_PEAK_GBS * 1e9), 4)}}
        del stack, feat
    out["note"] = ("frac = SURVEY 8(d)'s ALGORITHMIC work of the pass (all 15.36 M (point, triangle) pairs x 60 and 18 M (point, point) "
                   "pairs x 8 lane-ops per mesh; executed flops for the products) x meshes/s over the roof of its dominant kernel (78.6 T "
                   "lane-op/s un-fused fp32 VALU; 157.3 TFLOP/s fp32 MFMA).  Above 1 for config 3 at B = 8: the point-to-triangle scan "
                   "is CULLED (bounding spheres per 16 triangles: ~4 x fewer executed lane-ops than the brute-force count the survey "
                   "prices), so the algorithmic rate exceeds the issue roof -- the executed-instruction utilisation of that launch is "
                   "other_kernels.surface_scan (0.28-0.31).  At B = 1 every pass is a chain of launches of a few microseconds each on a "
                   "mostly idle chip")
    return out


def training_shape_times(dev, batch=16):
    """The shape the reference really trains at (GEOMetrics.py:25,44,50,66-68): batch 16, a 482-vertex / 960-face
    template with two 32-neighbour poles (meshgen.uv_sphere: same size and degree extremes as 482.obj), 3000 sampled
    vs 3000 gt points, one deformation block 1155 -> 192 x 13 -> 3 (14 0N-GCN layers + 13 vertex BatchNorms).  One
    "training-shape step" = block forward -> positions -> batch_point_to_surface -> backward -> Adam (56 tensors);
    HIP-graph replay bracketed by HIP events.  Reported beside the headline, not instead of it."""
    from geometrics_amd import models
    V, Fc = meshgen.uv_sphere()
    nv = V.shape[0]
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    faces = to(Fc)
    info = utils.adj_init(faces)
    base = to(meshgen.jittered_batch(V, batch, first=500))
    gt = to(meshgen.gt_cloud(batch, G_PTS, first=500))
    torch.manual_seed(3041)
    block = models.BatchMeshDeformationBlock(1155, nv).to(dev).train()
    feats = torch.randn(batch, nv, 3, device=dev, requires_grad=True)
    pooled = torch.randn(batch, nv, 1152, device=dev, requires_grad=True)
    params = list(block.parameters())
    opt = optim.FusedAdam(params, lr=1e-4)
    csr = layers.adjacency_csr(info["adj"])
    gt_index = ops.GtIndex(gt) if CULLED_CHAMFER else None      # static per gt cloud: the culled Chamfer tiles, as in the headline step

    def step():
        opt.zero_grad()
        feats.grad = pooled.grad = None
        with layers.deferred_parameter_gradients():
            f, coords = block(feats, pooled, info["adj"])
            loss = utils.batch_point_to_surface(base + coords, info, gt, num=S_PTS, gt_index=gt_index)
            loss.backward()
        opt.step()

    def block_only():
        opt.zero_grad()
        feats.grad = pooled.grad = None
        with layers.deferred_parameter_gradients():
            f, coords = block(feats, pooled, info["adj"])
            (f.sum() + coords.sum()).backward()

    pos = base.clone().requires_grad_(True)

    def loss_only():
        pos.grad = None
        utils.batch_point_to_surface(pos, info, gt, num=S_PTS, gt_index=gt_index).backward()

    t_step = event_time_us(step, iters=5, warm=3)
    t_block = event_time_us(block_only, iters=5, warm=2)
    t_loss = event_time_us(loss_only, iters=10, warm=2)
    return {"workload": "reference training shape: batch %d, %d vertices / %d faces (rows of up to %d adjacency entries), "
                        "deformation block 1155-192x13-3 fwd+bwd + surface loss (3000 vs 3000 points) + Adam"
                        % (batch, nv, Fc.shape[0], int((csr.rowptr[1:] - csr.rowptr[:-1]).max())),
            "step_us": round(t_step, 1), "meshes_per_s": round(batch / (t_step * 1e-6), 1),
            "deformation_block_fwd_bwd_us": round(t_block, 1), "surface_loss_fwd_bwd_us": round(t_loss, 1),
            "aggregation_kernel": "ELL table width %d%s" % (csr.ell_w, " + CSR tail for the long rows" if csr.over else "")
                                  if csr.ell_w else "generic CSR"}


def driver_step_times(dev, batch=16, profile_replays=0, zero_edit=False, overlap_losses=None):
    """The step the reference's driver really runs (GEOMetrics.py:110-174) at its own sizes: batch 16 on the 482-vertex /
    960-face template (meshgen.uv_sphere: the size and the two 32-neighbour poles of 482.obj), three poolings from four
    feature maps each (64x56^2, 128x28^2, 256x14^2, 512x7^2 -- what the three VGG encoders return; they are inputs here: the
    encoders are the reference's torch/MIOpen code and out of scope), three deformation blocks 963 / 1155 / 1155 -> 192 x 13
    -> 3, three surface losses (3000 sampled vs 3000 gt points), the edge and Laplacian terms, backward to every parameter
    and feature map, Adam over all ~170 parameter tensors.  HIP-graph replay bracketed by HIP events.  The F1 bookkeeping
    of GEOMetrics.py:137 (monitoring, synchronises the host) is not part of the captured step.
    profile_replays > 0: only replay the captured step that many times (tools/profile_driver_step.py under rocprofv3).
    zero_edit: what an UNMODIFIED GEOMetrics.py gets through overlay/ -- the overlay's hot-path names ARE these functions and
    classes (tests/test_overlay.py asserts identity), called the way the driver calls them and with nothing it does not have:
    eager launches from python, torch.optim.Adam, no gt_index, no deferred_parameter_gradients, no HIP graph, the f1=True
    bookkeeping of the third loss and the four .item() reads of its progress message (host synchronisations) included;
    wall clock over whole steps."""
    from geometrics_amd import models
    V, Fc = meshgen.uv_sphere()
    nv = V.shape[0]
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    faces = to(Fc)
    info = utils.adj_init(faces)
    initial = to(V)
    gt = to(meshgen.gt_cloud(batch, G_PTS, first=700))
    torch.manual_seed(3041)
    blocks = [models.BatchMeshDeformationBlock(c, nv).to(dev).train() for c in (963, 1155, 1155)]
    maps = [[torch.randn(batch, c, d, d, device=dev, requires_grad=True) for c, d in ((64, 56), (128, 28), (256, 14), (512, 7))]
            for _ in range(3)]
    img_info = torch.tensor([[30.0 + 10 * i, 25.0, 1.1] for i in range(batch)], device=dev)
    params = [p for m in blocks for p in m.parameters()]
    opt = optim.FusedAdam(params, lr=1e-4)
    gt_index = ops.GtIndex(gt) if CULLED_CHAMFER else None
    lap = utils.batch_get_lap_info

    # the driver clones positions and camera parameters in front of every call (GEOMetrics.py:118-137: in-place safety for its
    # own utils); these operators never write their inputs, so only the zero-edit step keeps the copies
    clone = (lambda t: t.clone()) if zero_edit else (lambda t: t)

    # the surface loss of a stage needs only that stage's positions: issued on a second stream as soon as they exist, its scan
    # (VALU-issue bound: brute-force Chamfer pairs) runs beside the next block's layer launches (latency / matrix-core bound);
    # autograd runs a node's backward on its forward's stream, so the gathers of the backward overlap likewise.  All three
    # losses on ONE side stream: their draws advance the same generator state, in stage order.
    overlap = DRIVER_STEP_OVERLAP if overlap_losses is None else bool(overlap_losses)
    overlap = overlap and not zero_edit
    loss_stream = torch.cuda.Stream(device=dev) if overlap else None

    stage_weights = torch.tensor([3 * .2] * batch + [3 * .2] * batch + [3 * 2.0] * batch, dtype=torch.float32, device=dev)

    def surface_term(p, wgt):
        surf = lambda: utils.batch_point_to_surface(p, info, gt, num=S_PTS, gt_index=gt_index, weight=wgt)
        if loss_stream is None:
            return surf()
        loss_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(loss_stream):
            return surf()

    # (the template expanded over the batch is a constant of the run: the edited driver materialises it once, the zero-edit
    # step keeps GEOMetrics.py:116's expand in front of every step)
    base_const = initial.unsqueeze(0).expand(batch, nv, 3).contiguous()

    def predict():
        base = initial.unsqueeze(0).expand(batch, nv, 3) if zero_edit else base_const
        # (the edited driver tells the pooling how many columns will be concatenated in front of its features -- 3 coordinates,
        # + the 192 previous features -- and uses utils.concat_features: no torch.cat of the 35 MB input, no slicing copies of its
        # gradient; the zero-edit step keeps torch.cat)
        cat = (lambda a, b_: torch.cat((a, b_), dim=-1)) if zero_edit else utils.concat_features
        room = (lambda n: 0) if zero_edit else (lambda n: n)
        # a stage's positions have six consumers (pooling, next block, next stage's sum, surface loss, two regularisers):
        # utils.fan_out hands each its own handle and sums their six gradients in one launch instead of five
        fan = (lambda t, n: (t,) * n) if zero_edit else utils.fan_out
        # (the cameras of the three poolings are the same: the edited driver forms them once)
        cam = img_info if zero_edit else utils.batch_camera_info(img_info)
        # ... and hands the pooling what will stand in front of its features (the coordinates, the previous features): the
        # pooling launch copies them into place, the concatenations below find them there
        fr = (lambda *t: None) if zero_edit else (lambda *t: t)
        f = utils.batched_pooling(maps[0], base, clone(cam), headroom=room(3), fronts=fr(base))
        f, p1 = blocks[0](base, f, info["adj"])
        p1 = fan(base + p1, 6)
        stacked = DRIVER_STEP_STACKED_LOSSES and not zero_edit and loss_stream is None
        s1 = None if zero_edit or stacked else surface_term(p1[3], .2)
        f = cat(f, utils.batched_pooling(maps[1], clone(p1[0]), clone(cam), headroom=room(3 + HID), fronts=fr(p1[1], f)))
        f, p2 = blocks[1](clone(p1[1]), f, info["adj"])
        p2 = fan(p2 + p1[2], 6)
        s2 = None if zero_edit or stacked else surface_term(p2[3], .2)
        f = cat(f, utils.batched_pooling(maps[2], clone(p2[0]), clone(cam), headroom=room(3 + HID), fronts=fr(p2[1], f)))
        _, p3 = blocks[2](clone(p2[1]), f, info["adj"])
        p3 = fan(p3 + p2[2], 2)
        if stacked:
            # the three stages' surface losses (GEOMetrics.py:134-138: weights .2 / .2 / 2, one ground truth) as ONE call on the 48
            # stacked meshes with per-mesh factors -- the loss is a mean over the batch, so stage s enters with 3 x its weight:
            # one draw / scan / finalize / gather launch instead of three each (the scan's pairs are the same; the fixed costs
            # of the launches and the 17-workgroup finalize passes are not)
            s3 = utils.batch_point_to_surface(torch.cat((p1[3], p2[3], p3[0])), info, torch.cat((gt, gt, gt)), num=S_PTS,
                                              weight=stage_weights)
        else:
            s3 = None if zero_edit else surface_term(p3[0], 2.0)
        if zero_edit:
            return p1[3:], p2[3:], p3
        if loss_stream is not None:
            torch.cuda.current_stream().wait_stream(loss_stream)
        return (s1,) + tuple(p1[4:]), (s2,) + tuple(p2[4:]), (s3,) + tuple(p3[1:])

    def losses(p1, p2, p3):
        # GEOMetrics.py:134-161 with its weights folded into the operators: surface_loss_k * (.2, .2, 2) (formed in predict(),
        # behind each stage); per stage 300 * edge(p_k) + .2 * (1500 * lap term [* .3 for stage 1] + 100 * displacement term)
        # as ONE node per stage (utils.stage_regularisers; the zero-edit step below keeps the driver's own expressions)
        # p1, p2: (surface loss, regulariser handle as `cur`, regulariser handle as `prev`); p3: (surface loss, cur)
        return utils.sum_losses(
            *[t for t in (p1[0], p2[0], p3[0]) if t is not None],
            utils.stage_regularisers(initial, p1[1], info, lap_weight=.2 * .3 * 1500, edge_weight=300),
            utils.stage_regularisers(p1[2], p2[1], info, lap_weight=.2 * 1500, move_weight=.2 * 100, edge_weight=300),
            utils.stage_regularisers(p2[2], p3[1], info, lap_weight=.2 * 1500, move_weight=.2 * 100, edge_weight=300))

    def zero():
        opt.zero_grad()
        for group in maps:
            for m in group:
                m.grad = None

    last = {}

    def step():
        zero()
        with layers.deferred_parameter_gradients():
            loss = losses(*predict())
            loss.backward()
        opt.step()
        last["loss"] = loss.detach()

    if zero_edit:
        adam = torch.optim.Adam(params, lr=1e-4)          # GEOMetrics.py:73
        p2s = utils.batch_point_to_surface

        def driver_step():
            adam.zero_grad()
            p1, p2, p3 = (h[0] for h in predict())
            s1 = p2s(p1.clone(), info, gt, num=S_PTS)
            s2 = p2s(p2.clone(), info, gt, num=S_PTS)
            s3, f1 = p2s(p3.clone(), info, gt, num=S_PTS, f1=True)
            surface = s1 * .2 + s2 * .2 + s3 * 2
            edge = utils.batch_calc_edge(p1.clone(), info) * 300
            edge += utils.batch_calc_edge(p2.clone(), info) * 300
            edge += utils.batch_calc_edge(p3.clone(), info) * 300
            l1 = torch.mean(torch.sum((lap(initial, info) - lap(p1, info)) ** 2, 2)) * 1500
            l2 = torch.mean(torch.sum((lap(p1, info) - lap(p2, info)) ** 2, 2)) * 1500
            l2 += torch.mean(torch.sum((p1 - p2) ** 2, 2)) * 100
            l3 = torch.mean(torch.sum((lap(p2, info) - lap(p3, info)) ** 2, 2)) * 1500
            l3 += torch.mean(torch.sum((p2 - p3) ** 2, 2)) * 100
            lap_loss = .2 * (l1 * .3 + l2 + l3)
            loss = edge + surface + lap_loss
            loss.backward()
            adam.step()
            return loss.item(), surface.item(), edge.item(), lap_loss.item(), float(f1)     # the driver's progress message
        for _ in range(3):
            driver_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            out = driver_step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        return {"workload": "the same step through the overlay's public names only, as an unmodified GEOMetrics.py:110-174 issues it: "
                            "eager launches, torch.optim.Adam, no gt_index / deferral / HIP graph, f1=True + the four .item() reads per step",
                "ms_per_step": round(dt * 1e3, 4), "meshes_per_s": round(batch / dt, 1), "final_loss": round(out[0], 5), "f1": round(out[4], 3),
                "timing": "wall clock over %d whole steps (host launch overhead and synchronisations are the point)" % n}

    if profile_replays:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            step()
        for _ in range(profile_replays):
            g.replay()
        torch.cuda.synchronize()
        return {"final_loss": float(last["loss"])}

    t_step = event_time_us(step, iters=5, warm=3)
    final = float(last["loss"])
    # stages in isolation (forward + backward of each on the step's own tensors; they do not add up to the step: the step's
    # stages share launches -- one end-of-pass reduction, one Adam launch per 64 tensors)
    base = initial.unsqueeze(0).expand(batch, nv, 3).contiguous()
    pos = (base + 0.01 * torch.randn_like(base)).requires_grad_(True)

    def pool_fb():
        zero()
        pos.grad = None
        utils.batched_pooling(maps[1], pos, img_info.clone()).sum().backward()
    feats = [torch.randn(batch, nv, c, device=dev, requires_grad=True) for c in (960, 1152)]   # + the 3 coordinates the block prepends

    def block_fb(i):
        def run():
            zero()
            feats[min(i, 1)].grad = None
            with layers.deferred_parameter_gradients():
                f, c = blocks[i](base, feats[min(i, 1)], info["adj"])
                (f.sum() + c.sum()).backward()
        return run

    def surf_fb():
        pos.grad = None
        utils.batch_point_to_surface(pos, info, gt, num=S_PTS, gt_index=gt_index).backward()

    def reg_fb():
        pos.grad = None
        utils.stage_regularisers(initial, pos, info, lap_weight=1500, edge_weight=300).backward()
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    stages = {"one batched_pooling (4 maps) fwd+bwd": event_time_us(pool_fb, iters=5, warm=2),
              "deformation block 963-192x13-3 fwd+bwd": event_time_us(block_fb(0), iters=5, warm=2),
              "deformation block 1155-192x13-3 fwd+bwd": event_time_us(block_fb(1), iters=5, warm=2),
              "one surface loss fwd+bwd": event_time_us(surf_fb, iters=10, warm=2),
              "edge + Laplacian terms of one stage fwd+bwd": event_time_us(reg_fb, iters=10, warm=2),
              "Adam over %d tensors" % len(params): event_time_us(lambda: opt.step(), iters=10, warm=2)}
    return {"workload": "the driver's own step (GEOMetrics.py:110-174): batch %d, %d vertices / %d faces, 3 x pooling of 4 feature maps, "
                        "deformation blocks 963 / 1155 / 1155 -> 192 x 13 -> 3, 3 surface losses (3000 vs 3000 points), edge + "
                        "Laplacian terms, backward, Adam over %d tensors; HIP-graph replay; image encoders and the F1 "
                        "bookkeeping not included" % (batch, nv, Fc.shape[0], len(params)),
            "ms_per_step": round(t_step / 1e3, 4), "meshes_per_s": round(batch / (t_step * 1e-6), 1), "final_loss": round(final, 5),
            "stages_us": {k: round(v, 1) for k, v in stages.items()}}


def whole_batch_times(dev, meshes=64, steps=20, warmup=5):
    """BASELINE config 5's WHOLE batch (64 meshes) on this one GPU, same step, HIP-graph replay: the strong-scaling anchor
    for the 8-GPU target (64 meshes / 8 GPUs = the 8-mesh shard that `value` is quoted on).  Reported beside the headline."""
    torch.cuda.empty_cache()
    w = Workload(dev, 0, meshes)
    w.capture()
    for _ in range(warmup):
        w.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        w.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"workload": "config 5 whole batch on ONE GPU: %d meshes per step (strong-scaling anchor; the headline is the "
                       "8-mesh weak-scaling shard)" % meshes,
           "ms_per_step": round(dt * 1e3, 4), "meshes_per_s": round(meshes / dt, 1), "final_loss": round(w.mean_loss(), 6)}
    del w
    torch.cuda.empty_cache()
    return out


def strong_scaling_point(dev, rank, world, force_dp, steps, warmup, global_batch=64):
    """N > 1, run by EVERY rank behind the weak-scaling region: BASELINE config 5's whole batch (64 meshes) split over the
    ranks (64 / N per GPU), same step, same timing rules (barrier + synchronize on both sides, MAX over ranks) -- so that
    one `--gpus N` run prints the weak AND the strong number.  At N = 1 the same figure is `whole_batch_single_gpu`."""
    first, count = gdist.shard_range(global_batch, rank, world)
    torch.cuda.empty_cache()
    w = Workload(dev, first, count, force_dp=force_dp)
    w.capture()
    gdist.barrier()
    elapsed = time_steps(w.run, steps, warmup)
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())
    w.finish()
    out = {"global_batch": global_batch, "meshes_per_gpu": count, "scaling": "strong", "ms_per_step": round(elapsed / steps * 1e3, 4),
           "meshes_per_s": round(global_batch * steps / elapsed, 1), "final_loss": round(w.mean_loss(), 6)}
    del w
    torch.cuda.empty_cache()
    return out


def split_bf16_experiment(dev, rows=20496, k=FEAT):
    """EXPERIMENT, on no route of the step (round-5 review item 5): the first layer's forward product with EXACT fp32 products on
    the bf16 matrix cores (every fp32 operand = the exact sum of three bf16 numbers; six exact a_i b_j products per element pair,
    fp32 accumulation: csrc/dense_split_bf16.hip) beside the library's and this package's fp32 matrix-core product on the same
    operands (rotated over three buffers: a step does not find its X in the infinity cache either); errors against float64 on
    a 2048-row sample.  dtype of the experiment: bf16 x 3 operands, fp32 accumulate -- the step's dtype stays f32."""
    from geometrics_amd import dense
    xs = [torch.randn(rows, k, device=dev) for _ in range(3)]
    w = torch.randn(k, HID, device=dev) * 0.05
    planes = dense.split_bf16_planes(w)
    outs = [torch.empty(rows, HID, device=dev) for _ in range(3)]
    turn = [0]

    def rotating(fn):
        def call():
            i = turn[0] = (turn[0] + 1) % 3
            fn(xs[i], outs[i])
        return call
    t_lib = event_time_us(rotating(lambda x, o: torch.mm(x, w, out=o)), iters=30)
    t_own = event_time_us(rotating(lambda x, o: dense.forward(x, w, out=o)), iters=30)
    t_s6 = event_time_us(rotating(lambda x, o: dense.gemm_split_bf16(x, planes, 6, out=o)), iters=30)
    t_s9 = event_time_us(rotating(lambda x, o: dense.gemm_split_bf16(x, planes, 9, out=o)), iters=30)
    a64, w64 = xs[0][:2048].double().cpu(), w.double().cpu()
    exact = a64 @ w64
    rms = lambda c: float((c[:2048].double().cpu() - exact).pow(2).mean().sqrt())
    e_split, e_native = rms(dense.gemm_split_bf16(xs[0], planes, 6)), rms(dense.forward(xs[0], w))
    fl = 2.0 * rows * k * HID
    return {"product": "[%d, %d] x [%d, %d] forward (layer 1), fp32 in / fp32 out" % (rows, k, k, HID),
            "dtype": "bf16 x 3-way split operands (exact products), fp32 accumulate -- experiment only, the step computes in f32 MFMA",
            "us": {"library fp32 (selection in use)": round(t_lib, 1), "fp32 matrix cores (dense_gemm.hip)": round(t_own, 1),
                   "split bf16, 6 terms": round(t_s6, 1), "split bf16, 9 terms": round(t_s9, 1)},
            "fp32_equivalent_tflops_split6": round(fl / t_s6 / 1e6, 1),
            "rms_error_vs_float64": {"split bf16 (6 terms)": e_split, "native fp32 MFMA": e_native, "ratio": round(e_split / e_native, 3)},
            "verdict": "more accurate than the native fp32 matrix-core chain (the leading accumulator is rounded 31 times instead of 241) "
                       "and NOT faster at this shape: 6 -> 9 terms (+50 % MFMAs) costs ~9 us of ~86, the launch is bound by operand "
                       "delivery (every workgroup re-reads the 1.1 MB of weight planes through L2, X is split in registers), not by "
                       "matrix-core issue; not promoted"}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(budget_s=12.0):
    """Reference-side CPU throughput for the part of the path that has a CPU implementation:
    Chamfer NN (the reference's own nnsearch when oracle/_ref is present) + the C restatement
    of tri_distance, single thread as shipped, on whole meshes of the same workload."""
    import oracle
    oracle.build()
    V, Fc = meshgen.icosphere(V_LEVEL)
    use_ref = oracle.have_ref()
    done, t_nn, t_tri = 0, 0.0, 0.0
    t_start = time.perf_counter()
    while done < 2 or (time.perf_counter() - t_start < budget_s and done < 64):
        verts = meshgen.jittered_batch(V, 1, first=done)
        gt = meshgen.gt_cloud(1, G_PTS, first=done)
        pred = meshgen.gt_cloud(1, S_PTS, first=1000 + done)
        t0 = time.perf_counter()
        oracle.chamfer_nn(gt, pred, use_ref=use_ref)
        t1 = time.perf_counter()
        oracle.tri_scan_indexed(gt, verts, Fc)
        t2 = time.perf_counter()
        t_nn += t1 - t0
        t_tri += t2 - t1
        done += 1
    # all-cores figure: one independent mesh per thread (the reference has no threading; ctypes releases the GIL)
    import concurrent.futures
    threads = os.cpu_count() or 1      # every host CPU (round 4 capped this at 64 and still called it "all cores")
    def one_mesh(i):
        verts = meshgen.jittered_batch(V, 1, first=i)
        gt = meshgen.gt_cloud(1, G_PTS, first=i)
        pred = meshgen.gt_cloud(1, S_PTS, first=1000 + i)
        oracle.chamfer_nn(gt, pred, use_ref=use_ref)
        oracle.tri_scan_indexed(gt, verts, Fc)
    with concurrent.futures.ThreadPoolExecutor(threads) as ex:
        t0 = time.perf_counter()
        list(ex.map(one_mesh, range(threads)))
        t_all = time.perf_counter() - t0
    return {"value": round(done / (t_nn + t_tri), 3), "unit": "meshes/s (Chamfer NN both directions + tri_distance only)",
            "all_cores_value": round(threads / t_all, 2), "all_cores_threads": threads, "cpu_model": _cpu_model(),
            "cores": 1, "kind": "port",   # the combined figure is dominated by tri_distance, which only exists as our C port
            "chamfer_kind": "reference" if use_ref else "port",
            "sample": "%d whole meshes of the bench workload (3000x3000 NN both ways, 3000 pts x 5120 faces); "
                      "NN = %s, tri_distance = C restatement (the reference has no CPU tri_distance)"
                      % (done, "reference nnsearch (oracle/_ref)" if use_ref else "C restatement of nnsearch"),
            "chamfer_only_meshes_per_s": round(done / t_nn, 2), "tri_only_meshes_per_s": round(done / t_tri, 3),
            "host_cpus": os.cpu_count()}


def parity_spot_check(w):
    """OUTSIDE the timed region: one evaluation of the surface loss on the route the step times (visiting-order draws under the
    gt index, culled Chamfer tiles, finalize roles in the scan launch) at the positions the timed steps ended on, checked
    against the CPU oracle on the very (choices, u, v) the generator drew: mismatching NN / triangle indices and region codes
    (bit-exact is the bar: 0), arg-min distances whose bits differ, the loss against the CPU restatement of utils.py:441-502
    (1e-5) and the gradient with respect to the positions against its fp32 autograd (max-norm, 1e-4)."""
    import oracle
    from oracle import ref_ops
    oracle.build()
    pos = w.positions().detach().clone().requires_grad_(True)
    seen = {}
    ops.scan_capture = seen
    try:
        loss = utils.batch_point_to_surface(pos, w.info, w.gt, num=S_PTS, gt_index=w.gt_index)
    finally:
        ops.scan_capture = None
    loss.backward()
    torch.cuda.synchronize()
    cpu = lambda t: t.detach().cpu().numpy()
    verts, gt, faces = cpu(pos), cpu(w.gt), cpu(w.faces)
    ch, u, v = cpu(seen["choices"]), cpu(seen["u"]), cpu(seen["v"])
    pred = ref_ops.sample_points(*(torch.from_numpy(a) for a in (verts, faces, ch, u, v))).numpy()
    d_gt, i_gt, d_pred, i_pred = oracle.chamfer_nn(gt, pred)
    t_d, t_opt, t_idx = oracle.tri_scan_indexed(gt, verts, faces)
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    cv = torch.from_numpy(verts).requires_grad_(True)
    ref = ref_ops.point_to_surface(cv, torch.from_numpy(faces), torch.from_numpy(gt), torch.from_numpy(ch), torch.from_numpy(u),
                                   torch.from_numpy(v))
    ref.backward()
    g, rg = cpu(pos.grad), cv.grad.numpy()
    return {"meshes": int(verts.shape[0]), "route": "%s, finalize roles %s"
                                                     % ("gt_index (culled Chamfer tiles, visiting-order draws)" if w.gt_index is not None
                                                        else "brute-force Chamfer tiles, independent draws", "on" if ops.scan_finalize_tail else "off"),
            "idx_mismatches": int((cpu(seen["idx_gt"]) != i_gt).sum() + (cpu(seen["idx_pred"]) != i_pred).sum()
                                  + (cpu(seen["tri_index"]) != t_idx).sum() + (cpu(seen["tri_option"]) != t_opt).sum()),
            "sampled_point_bit_mismatches": int((bits(cpu(seen["points"])) != bits(pred)).sum()),
            "dist_bit_mismatches": int((bits(cpu(seen["sq_pred"])) != bits(d_pred)).sum() + (bits(cpu(seen["tri_dist"])) != bits(t_d)).sum()),
            "loss": float(loss.item()), "loss_cpu_restatement": float(ref.item()),
            "loss_rel_err": float(abs(loss.item() - ref.item()) / abs(ref.item())),
            "grad_pos_max_err_over_scale": float(np.abs(g - rg).max() / np.abs(rg).max())}


def self_launch(n, argv):
    """`python bench.py --gpus N` typed as is (no torchrun, WORLD_SIZE unset): this process becomes the launcher -- it starts
    N copies of this script, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT in
    their environment, exactly what torch.distributed.run would set), passes rank 0's single JSON line through on stdout and
    returns the worst exit code.  A rank that dies takes the others down with it (their PIDs, nothing by pattern)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        alive = set(range(n))
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0:
                    rc = rc or code
                    print("bench.py: rank %d exited with code %d; stopping the other ranks" % (r, code), file=sys.stderr)
                    for q in alive:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--meshes-per-gpu", type=int, default=8)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="STRONG scaling: this many meshes in total, split evenly over the ranks (64 = BASELINE config 5's batch); "
                         "the line then says scaling: strong.  Default 0: weak scaling, --meshes-per-gpu on every rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--clock-warmup-ms", type=float, default=250.0,
                    help="neutral GEMM load in front of the warm-up steps so that short runs are not timed on the clock ramp (0: off)")
    ap.add_argument("--gt-index", action="store_true", help="culled Chamfer tiles on an index of the ground-truth clouds built OUTSIDE the "
                    "step (static clouds only; same results bit for bit).  Default: brute-force tiles, everything that depends on the "
                    "ground truth is inside the step")
    ap.add_argument("--plain-chamfer", action="store_true", help="(the default since round 6; kept for older command lines)")
    ap.add_argument("--separate-finalize", action="store_true", help="the surface loss's finalize pass as a launch of its own instead of "
                    "extra workgroups of the scan launch (same results; the A/B switch of ops.scan_finalize_tail)")
    ap.add_argument("--launch", choices=("graph", "eager"), default="graph",
                    help="replay the whole step as one HIP graph (default) or launch eagerly from python")
    ap.add_argument("--breakdown", action="store_true", help="also print a per-stage event-timed breakdown")
    ap.add_argument("--steps-only", action="store_true",
                    help="no per-kernel timing loops (they capture HIP graphs, which rocprofv3 --pmc cannot trace): "
                         "what tools/pmc_traffic.sh runs together with --launch eager")
    args = ap.parse_args()
    global CULLED_CHAMFER
    CULLED_CHAMFER = bool(args.gt_index) and not args.plain_chamfer
    if args.separate_finalize:
        ops.scan_finalize_tail = False

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    shared_gpu = bool(os.environ.get("GEOM_DIST_BACKEND"))     # test hook: ranks share a device, gloo carries the collective
    if args.gpus > torch.cuda.device_count() and not shared_gpu:
        raise SystemExit("--gpus %d but only %d HIP device(s) are visible (one rank per GPU)" % (args.gpus, torch.cuda.device_count()))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # typed without a launcher: start the N ranks ourselves
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank, world, local = gdist.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher started another number of ranks" % (args.gpus, world))
    # GEOM_BENCH_FORCE_DP=1 (test hook, N = 1 only): run the WHOLE N > 1 code path of this file -- RCCL process group, bucket,
    # two-graph step, async all-reduce, the probes with RCCL's threads alive -- in a 1-rank group on one GPU
    force_dp = world == 1 and os.environ.get("GEOM_BENCH_FORCE_DP", "0") not in ("", "0")
    if force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        torch.cuda.set_device(local % torch.cuda.device_count())
        torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1)
    dev = torch.device("cuda", local % torch.cuda.device_count())   # one rank per GPU (modulo only for 1-GPU tests)
    torch.cuda.set_device(dev)

    tuned = gemm_tuning.enable()      # pin the measured-fastest library GEMM per shape (no tuning at run time)
    per_gpu = args.meshes_per_gpu
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit("--global-batch %d does not split evenly over %d ranks" % (args.global_batch, world))
        per_gpu = args.global_batch // world
    first, count = gdist.shard_range(per_gpu * world, rank, world)
    w = Workload(dev, first, count, force_dp=force_dp)
    if not tuned and gemm_tuning.status == "library default (tuning file rejected)" and os.environ.get("GEOM_RETUNE", "1") != "0":
        # the shipped selections belong to another library build: pick this build's once (~1 s), instead of running every
        # library product of the run on the default heuristic
        gemm_tuning.tune_products([(count, w.nv, FEAT, HID), (count, w.nv, HID, HID)], dev)
    launch = "eager"
    if args.launch == "graph":
        try:
            w.capture()
            launch = "hipgraph"
        except Exception as exc:  # fall back loudly, never silently
            print("bench.py: HIP graph capture failed (%s: %s); running eager" % (type(exc).__name__, exc),
                  file=sys.stderr)
            w.graphs = None
            torch.cuda.synchronize()
            from geometrics_amd import _lib
            _lib.clear_hip_error()
    gdist.barrier()          # ranks finish their setup at different times: condition the devices together, so that
    # the same K steps behind the same W warm-up steps WITHOUT the conditioning load first (what --clock-warmup-ms 0 prints):
    # reported beside the headline as ms_per_step_unconditioned
    unconditioned = time_steps(w.run, args.steps, args.warmup) if args.clock_warmup_ms > 0 else None
    settle_clocks(dev, args.clock_warmup_ms)   # nobody idles at the barrier in front of the timed region and cools down again
    elapsed = time_steps(w.run, args.steps, args.warmup)
    if unconditioned is None:
        unconditioned = elapsed
    if world > 1:
        tu = torch.tensor([unconditioned], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tu, op=torch.distributed.ReduceOp.MAX)
        unconditioned = float(tu.item())
    # [max, -min] of the ranks' own clocks in one MAX all-reduce; ranks_seen = an all-reduce of ones through the SAME group the
    # gradient bucket uses, so the line shows how many ranks the collective really spanned
    own = elapsed
    t = torch.tensor([elapsed, -elapsed], device=dev, dtype=torch.float64)
    seen = torch.ones((), device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(seen, op=torch.distributed.ReduceOp.SUM)
    elapsed, fastest = float(t[0].item()), -float(t[1].item())
    ranks_seen = int(round(float(seen.item())))
    strong = None
    if (world > 1 or force_dp) and not args.global_batch and 64 % world == 0 and not args.steps_only:
        try:
            strong = strong_scaling_point(dev, rank, world, force_dp, args.steps, args.warmup)
        except Exception as exc:          # every rank fails or none does (same code, same sizes): the headline still goes out
            print("bench.py: strong-scaling point failed on rank %d (%s: %s)" % (rank, type(exc).__name__, exc), file=sys.stderr)
            strong = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}

    if rank == 0:
        total_meshes = per_gpu * world * args.steps
        line = {
            "metric": "meshes/sec (sample+Chamfer+tri_dist+0N-GCN fwd+bwd), 2562 verts/3000 pts",
            "value": round(total_meshes / elapsed, 2), "unit": "meshes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "ranks_seen": ranks_seen,
            "ms_per_step_unconditioned": round(unconditioned / args.steps * 1e3, 4),
            "ms_per_step_ranks": {"max": round(elapsed / args.steps * 1e3, 4), "min": round(fastest / args.steps * 1e3, 4),
                                  "rank0": round(own / args.steps * 1e3, 4)},
            "config": {"workload": "BASELINE config 5 shard: %d independent 2562-vert/5120-face meshes per GPU, full "
                                   "loss (face sampling 3000 pts + Chamfer NN vs 3000 GT pts + tri_distance 3000x5120 + "
                                   "point-to-surface) on top of a 3-layer 0N-GCN 963-192-192-192, fwd+bwd, %sAdam step"
                                   % (per_gpu, "flat-bucket grad all-reduce over %d ranks (%s), " % (world, torch.distributed.get_backend())
                                      if world > 1 else "single process (no collective), "),
                       "meshes_per_gpu": per_gpu, "global_batch": per_gpu * world, "parallelism": "dp%d" % world,
                       "launch": launch, "gemm_selection": gemm_tuning.status,
                       "clock_warmup_ms": args.clock_warmup_ms,
                       "dp_sequence": None if world == 1 and not force_dp else
                       ("per step ONE graph: [Adam on the bucket the previous step all-reduced, forward, backward, reduction launch "
                        "-> bucket, the all-reduce (gradients + loss, 1.04 MB) on RCCL's stream inside the capture, the first layer's "
                        "input gradient beside it, join]" if w.dp_sequence == "captured" else
                        "per step: graph A [Adam on the bucket the previous step all-reduced, forward, backward, reduction launch "
                        "-> bucket] ; ONE async all-reduce (gradients + loss, 1.04 MB) beside graph B [first layer's input "
                        "gradient] ; the launch stream waits for the collective")},
            "final_loss": round(w.mean_loss(), 6),
        }
        if strong is not None:
            line["strong_scaling"] = strong
        if per_gpu == 8:
            line["step_level"] = step_level_roofline(total_meshes / elapsed / world)     # per GPU
        if not args.steps_only:
            try:
                roofline, others = kernel_rooflines(w)
                line["roofline"] = roofline
                line["other_kernels"] = others
            except Exception as exc:       # the headline must reach the driver whatever happens to the per-kernel probes
                print("bench.py: per-kernel probes failed (%s: %s); the line goes out without them" % (type(exc).__name__, exc),
                      file=sys.stderr)
                line["roofline"] = None
                line["roofline_error"] = "%s: %s" % (type(exc).__name__, str(exc)[:300])
                from geometrics_amd import _lib
                _lib.clear_hip_error()
        def extra(key, fn):       # a side measurement never takes the headline down with it
            try:
                line[key] = fn()
            except Exception as exc:
                print("bench.py: %s failed (%s: %s)" % (key, type(exc).__name__, exc), file=sys.stderr)
                line[key] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        if world == 1 and not force_dp and not args.steps_only:
            extra("components_us", lambda: component_times(w))
            extra("configs", lambda: baseline_configs(dev))
            extra("split_bf16_experiment", lambda: split_bf16_experiment(dev))
            extra("reference_training_shape", lambda: training_shape_times(dev))
            extra("driver_step", lambda: driver_step_times(dev))
            extra("driver_step_zero_edit", lambda: driver_step_times(dev, zero_edit=True))
            extra("whole_batch_single_gpu", lambda: whole_batch_times(dev))
            if isinstance(line.get("whole_batch_single_gpu"), dict) and "ms_per_step" in line["whole_batch_single_gpu"]:
                wb = line["whole_batch_single_gpu"]        # N = 1: the strong-scaling point IS the whole batch on this GPU
                line["strong_scaling"] = {"global_batch": 64, "meshes_per_gpu": 64, "scaling": "strong", "ms_per_step": wb["ms_per_step"],
                                          "meshes_per_s": wb["meshes_per_s"], "final_loss": wb["final_loss"]}
        if world == 1 and not args.no_cpu_baseline:
            extra("cpu_baseline", cpu_baseline)
            extra("parity_spot_check", lambda: parity_spot_check(w))
        if force_dp:
            line["config"]["forced_dp"] = "1-rank RCCL group: the N > 1 step sequence on one GPU (test hook GEOM_BENCH_FORCE_DP)"
        print(json.dumps(line))
    gdist.barrier()
    if world > 1 or force_dp:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
