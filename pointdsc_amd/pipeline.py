"""Several forwards of one ``PointDSC`` module in flight: consecutive batches alternate between ``depth`` HIP streams, each with
its own workspace, so that the latency-bound tail of one forward (NMS, seed ranking, kNN, per-seed solver, scoring,
refinement: a chain of ~15 small launches that occupy a few CUs) overlaps the compat build and the first layers of the next.

Measured on one MI355X (profiles/r03_b_overlap_probe.txt): 32 pairs of N=5000 per step 14.96 -> 14.57 ms, 4 pairs (the
per-GPU share of the 8-GPU configuration) 2.06 -> 1.89 ms, one pair of N=1000 0.558 -> 0.354 ms per step.  Results are bit
for bit those of the plain call: every forward is still one ``pdsc_forward_testing`` over its own workspace; only WHEN its
kernels run changes.  Weights (packed / split buffers) are shared and read-only.

    runner = InFlight(model, depth=2)
    for data in batches:                   # tensors on the GPU
        res = runner(data)                 # returns at once; res["ready"] is an event on the forward's stream
        ...
    runner.synchronize()                   # or, per result: torch.cuda.current_stream().wait_event(res["ready"]) (and
                                           # t.record_stream(current) for result tensors that outlive the runner's next use
                                           # of the slot) before using res on another stream

``tail_streams=True`` (default) enqueues each forward on TWO streams (``pdsc_forward_testing_streams``): the encoder on the
slot's stream, everything after it on a high-priority companion stream, so that the tail's few workgroups are dispatched ahead
of the queued workgroups of the other slot's attention launch.  Measured, interleaved in one process, final build
(profiles/r03_h_inflight_ab.txt, r03_i_inflight_ab.txt; ms per step: one stream / two plain streams / two streams + tail streams
/ three plain streams): 32 pairs of N=5000 15.08 / 15.06 / 14.69 / 14.71; 16 pairs 8.59 / 8.60 / 8.20 / 8.25; 8 pairs 4.46 / 4.48
/ 4.09 / 4.04; 4 pairs 2.05 / 2.06 / 1.93 / 1.96; KITTI 16 pairs 8.81 / 8.82 / 8.39 / 8.42, 2 pairs 1.44 / 1.46 / 1.23 / 1.15;
N=10000 8 pairs 14.92 / 14.95 / 14.57 / 14.60, 1 pair 2.33 / 2.37 / 2.04 / 1.95.  Two plain streams alone gain nothing with the
final attention kernel (they did, 1.4-5 %, while its workgroups still reserved 132 KiB of LDS: r03_f / r03_g); single pairs of
N=1000 lose 10 % with tail streams and take the hipGraph path below instead.  (Also tried and removed: holding forward i+1 back
until forward i's encoder is done so that the two overlap tail-on-head -- as slow as one stream, r03_g.)

``graphs=True`` additionally captures each slot's forward in a hipGraph (``pdsc_forward_testing`` only enqueues kernels: no
sync, no allocation) and replays it: the ~45 launches of a forward cost the host one call.  That only matters when a forward is
launch-bound -- one pair of N=1000: 0.243 ms per forward with three eager forwards in flight, 0.149 ms with four captured ones
(profiles/r03_e_graph_probe.txt); at 32 pairs of N=5000 it changes nothing.  Inputs are copied into per-slot static tensors,
outputs are returned as copies; ragged batches and the validation forward take the eager path.
``zero_copy=True`` (with graphs; r04) drops that staging: a slot's graph is captured ON the caller's fp32 input tensors and its own
output tensors are handed out -- for callers that keep feeding the same buffers (a serving loop with pre-allocated inputs; other
buffers are captured as a further graph, up to 4 per slot) and read a result before its slot runs again (``depth`` calls later).
Five small launches and two allocations less per forward: the host's share of a captured N = 1000 forward, where the host is the
bound (bench.py n1000_b1, profiles/r04_v_*).
Exactness under concurrency (r03; tools/inflight_race_probe.py, tools/inflight_diverge_probe.py, profiles/r03_*_probe.txt): with
several forwards sharing the chip -- above all three replayed graphs of 2-3 pairs of N=5000 -- up to 35 % of the forwards first
came back with a pose off by 1e-4 ... 6e-4 (labels equal).  Both causes sat in the hypothesis scoring stage and are fixed in
the library: (1) hipMemsetAsync of the vote counters was not reliably ordered before the atomicAdds that followed it on the
same stream (now: no memset, no atomics; csrc/score.hip, csrc/pdsc_common.h launch_fill_u32); (2) the compiler-vectorised
packed-fp32 form of the residual test (v_pk_fma_f32 with op_sel broadcasts) miscounted a few votes on one half of the seed pairs
while other kernels were co-resident -- inputs verified equal, a recount inside the same launch right, the affected half moving
with the instruction schedule -- and score.hip is now built without SLP vectorisation (pointdsc_amd/build.py).  After both:
0 differing forwards in 8000-12000 per mode (plain streams, tail streams, replayed graphs; 1, 2 and 3 pairs of N=5000, single
pairs of N=1000).

The reference has nothing like it (its testing loop is one synchronous call per pair, evaluation/test_3DMatch.py:32-54).
"""
from __future__ import annotations

import warnings
from typing import Callable, Dict, Optional

import torch


class InFlight:
    def __init__(self, model, depth: int = 2, device=None, graphs: bool = False, tail_streams: bool = True, zero_copy: bool = False):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.model = model
        self.depth = depth
        # graphs only: capture each slot's graph ON the caller's input tensors (fp32, contiguous) and hand out the graph's own
        # output tensors -- no staging copies in, no clones out: 5 small launches and two allocations less per forward, which is
        # most of the host's share of a captured N = 1000 forward.  Contract: the caller keeps feeding the same input buffers (new
        # contents in place are fine once the previous forward on them has finished; other buffers are captured as a further
        # graph, up to 4 per slot) and a result is valid until the same slot runs again, `depth` calls later.
        self.zero_copy = bool(zero_copy) and bool(graphs) and depth > 1
        dev = device if device is not None else next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("pointdsc_amd has no CPU path: move the model to the GPU first")
        self.device = dev
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(depth)] if depth > 1 else [None]
        # workspace slots of the module, private to this runner (slot 0 = the module's plain calls): two runners on one module
        # must never share a workspace, their streams are not ordered against each other
        if not hasattr(model, "_next_ws_slot"):
            model._next_ws_slot = 1
        self._slots = list(range(model._next_ws_slot, model._next_ws_slot + depth)) if depth > 1 else [0]
        if depth > 1:
            model._next_ws_slot += depth
        # Each slot's forward is enqueued on two streams (pdsc_forward_testing_streams): the encoder on the slot's stream, the
        # latency-bound tail (classifier ... refinement: ~15 small launches) on a HIGH-priority companion stream, so that its few
        # workgroups are dispatched ahead of the queued workgroups of the other slot's attention launch.
        self.tail_streams = bool(tail_streams) and depth > 1 and not graphs     # (captured forwards are launch-bound problems: one stream)
        if self.tail_streams:
            for k, st in zip(self._slots, self.streams):
                tail = torch.cuda.Stream(device=dev, priority=-1)
                fork, join = torch.cuda.Event(), torch.cuda.Event()
                fork.record(st)                        # (creates the underlying hipEvents: their handles are passed to the library)
                join.record(tail)
                model._tail[k] = (tail, fork, join)
            torch.cuda.synchronize(dev)
        self._i = 0
        self.graphs = bool(graphs) and depth > 1
        self._captured = {}                     # slot -> {key: (key, graph, static inputs, static outputs)}
        self._ensure_weights()

    def _ensure_weights(self) -> None:
        """The packed / split weight buffers are built lazily by the first forward after a weight change -- on ONE stream; the
        other streams must not read them before that build has finished."""
        m = self.model
        if m._wpack is None or (m.attention_precision != "fp32" and m._wsplit is None):
            with torch.cuda.device(self.device):
                m.packed_weights(self.device)
                if m.attention_precision != "fp32":
                    m.split_weights(self.device)
                torch.cuda.current_stream(self.device).synchronize()

    def __call__(self, data: Dict, post: Optional[Callable[[Dict], object]] = None) -> Dict:
        """Enqueue one forward (and ``post(res)``, e.g. the pose all_gather of a multi-GPU job, behind it on the same
        stream).  Inputs may have been produced on the caller's current stream: the forward's stream waits for it."""
        self._ensure_weights()
        slot = self._i % self.depth
        self._i += 1
        s = self.streams[slot]
        if s is None:                                            # depth 1: the plain synchronous-enqueue call
            with torch.no_grad():
                self.model._guard_override = "lazy" if self.model.range_guard != "off" else "off"      # (no host sync inside a pipeline)
                try:
                    res = self.model(data)
                finally:
                    self.model._guard_override = None
            if post is not None:
                res["post"] = post(res)
            return res
        s.wait_stream(torch.cuda.current_stream(self.device))
        # the inputs were allocated on the caller's stream: tell the caching allocator that stream `s` reads them, or a tensor
        # the caller drops right after this call could be handed out again while the forward is still running
        for v in data.values():
            for t in (v if isinstance(v, (list, tuple)) else (v,)):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(s)
        with torch.cuda.stream(s), torch.no_grad():
            res = self._replay(slot, s, data) if self.graphs else None
            if res is None:
                self.model._ws_slot = self._slots[slot]
                self.model._guard_override = "lazy" if self.model.range_guard != "off" else "off"
                try:
                    res = self.model(data)
                finally:
                    self.model._ws_slot = 0
                    self.model._guard_override = None
            if post is not None:
                res["post"] = post(res)
            ev = torch.cuda.Event()
            ev.record(s)
        res["ready"] = ev
        return res

    def _replay(self, slot: int, s, data: Dict):
        """Replay (capturing first, if needed) the slot's hipGraph of `model(data)`; None = not capturable, run eagerly."""
        m = self.model
        corr = data.get("corr_pos")
        if not (torch.is_tensor(corr) and "testing" in data and data.get("num_corr") is None):
            return None
        names = ("corr_pos", "src_keypts", "tgt_keypts")
        zc = self.zero_copy and all(torch.is_tensor(data.get(k)) and data[k].dtype == torch.float32 and data[k].is_contiguous() for k in names)
        key = (tuple(corr.shape), m.attention_precision, m.compat_format, m.layer_gemm, m._wpack_key,
               tuple(data[k].data_ptr() for k in names) if zc else None)
        caps = self._captured.setdefault(slot, {})
        cap = caps.get(key)
        if cap is None:
            try:
                # zero-copy: the graph reads the caller's tensors (kept referenced here, so their memory stays theirs)
                static = {k: (data[k].detach() if zc else data[k].detach().to(torch.float32).contiguous().clone()) for k in names}
                static["testing"] = True
                m._ws_slot = self._slots[slot]
                # (captured forwards: the library's NaN poses are the range guard -- nothing of the module's guard can run inside a graph)
                m._guard_override = "off"
                try:
                    for _ in range(2):                   # workspace, packed weights, H3 range check: all before the capture
                        m(static)
                    s.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=s):
                        out = m(static)
                finally:
                    m._ws_slot = 0
                    m._guard_override = None
                cap = (key, g, static, out)
                if len(caps) >= 4 or any(k[:5] != key[:5] for k in caps):      # other shape / arithmetic / weights: the old graphs are dead
                    caps.clear()
                caps[key] = cap
            except Exception as e:        # noqa: BLE001  (capture is an optimisation: fall back to the eager path for good)
                warnings.warn(f"pointdsc_amd.InFlight: hipGraph capture failed ({e!r}); continuing without graphs", RuntimeWarning)
                self.graphs = False
                return None
        _, g, static, out = cap
        if not zc:
            for k in names:
                if data[k].data_ptr() != static[k].data_ptr():
                    static[k].copy_(data[k], non_blocking=True)
        g.replay()
        if zc:
            return {"final_trans": out["final_trans"], "final_labels": out["final_labels"], "M": None}
        return {"final_trans": out["final_trans"].clone(), "final_labels": out["final_labels"].clone(), "M": None}

    def synchronize(self) -> None:
        for s in self.streams:
            if s is not None:
                s.synchronize()          # (every forward ends with its main stream waiting for its tail stream)
        if self.depth == 1:
            torch.cuda.current_stream(self.device).synchronize()
        self.model._poll_range(block=False)     # the range words of the finished forwards (model.range_guard, lazy inside a pipeline)

    def close(self) -> None:
        """Wait for the forwards in flight and release this runner's workspaces (2.6 GB each for 32 pairs of N = 5000)."""
        self.synchronize()
        self._captured.clear()
        if self.depth > 1:
            for k in self._slots:
                self.model._workspaces.pop(k, None)
                self.model._tail.pop(k, None)
