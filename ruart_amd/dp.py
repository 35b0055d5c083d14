"""Single-node data parallelism for the SDNet training step: one process per GPU, RCCL all-reduce over xGMI.

The reference has no distributed code at all; this is the new exchange step the north star
asks for.  Design, for 8 MI355X on a fully connected xGMI mesh:
  * every rank runs the whole step on its own B-sample shard (samples are independent; the whole-tensor layer
    norm couples samples only inside a replica, so the semantics are "N reference steps, gradients averaged");
  * only what must move moves: trainable, actually-used parameters.  The dead ``get_answer.rnn.*`` GRU (never gets a
    gradient, Models/Layers.py:395-397) is excluded.  The word-embedding rows >= tune_partial are re-pinned after every
    step (Models/SDNetTrainer.py:369-373) so their update never survives - but their gradients do enter the global clipping
    norm (:366), and every rank must clip by the same coefficient or the replicas drift.  Three ways to account for them:
      - default ("full"): the whole tables are exchanged like every other gradient, so the clip norm is EXACTLY the norm of the
        averaged gradient - the semantics the module promises ("N reference steps, gradients averaged").  Costs 2 x (V -
        tune_partial) x 300 x 4 bytes of extra payload (78 MB instead of 39 MB per step at the shipped vocabulary: ~0.2 ms on
        8 xGMI-connected GPUs, against a 20-27 ms step);
      - ``opt['dp_pinned_scalar']`` (needs the fused optimizer): the rows are NOT exchanged; each rank adds the squared norm of
        its own pinned-row gradients to ONE scalar that rides along with the buckets (``pinned_sq``) and the clip norm becomes
        sqrt(|averaged trained gradients|^2 + sum_r |g_r,pinned|^2 / world^2).  That equals the exact norm only when the ranks'
        batches touch disjoint pinned rows; when they share rows (frequent out-of-head words) the cross terms 2 <g_r, g_s> /
        world^2 are missing and the norm is UNDER-estimated by up to sqrt(world) - a larger clip coefficient than N averaged
        reference steps would use (tests/test_dp_gloo.py::test_clip_norm_when_ranks_share_pinned_rows shows both cases);
      - ``opt['dp_skip_pinned_rows']`` (older switch) exchanges the head rows and zeroes the rest, dropping them from the norm.
    A sparse exchange (ids + rows of the touched pinned rows) was considered and dropped: at the bench shape a rank looks up
    ~19 k words per step, i.e. most of a 20 k vocabulary - the "sparse" all-gather would move MORE than the dense all-reduce;
  * gradients are packed into large flat buckets, allocated once (on point-to-point xGMI links large messages win; there is
    no NVSwitch-style in-network reduction to amortise small ones).  DEFAULT: the exchange runs AFTER ``backward()``, on the
    caller's stream, as synchronous collectives - one multi-tensor copy into the bucket, one all-reduce, one scale, and every
    parameter's ``.grad`` is pointed at its slice of the bucket (no copy back).  A synchronous c10d collective is enqueued on the
    CURRENT stream, so the whole exchange costs no cross-stream event at all, and it is not dead time for the GPU: the next
    batch's frozen encoder keeps running on its own stream (bert.py) while RCCL moves the ~80 MB.  That is the overlap this step
    offers - it beats the textbook one below on this workload: measured on one MI355X under a world-size-1 RCCL group
    (tools/dp_overhead2.py, 64-sample step) 25.1 ms against 25.0 without DP, where the overlapped mode costs 28.1-29.5 ms;
  * ``opt['dp_overlap_backward']`` (the round-2 default): 16 MB buckets in reverse parameter order; when the last gradient of a
    bucket has been produced (post-accumulate hooks) its copy and an ASYNCHRONOUS all-reduce are enqueued on a communication
    stream, strictly in bucket order, overlapping the rest of backward; ``average_gradients`` waits.  The trunk runs its
    question / object / OCR branches on three streams and autograd joins them only at the END of backward, so every hook notes
    the stream its gradient was accumulated on and the communication stream waits for all of them.  Those waits are what is
    expensive here, not the payload: with the copy, the collective and the allocator bookkeeping all removed and ONLY the 14
    ``wait_stream`` calls per step left, the step is still 28.0 ms against 25.0 (an event hop itself is 17 us,
    tools/stream_hop.py; the rest is the branches' and the encoder stream's interleaving being disturbed - every variant that
    makes a second stream wait on a trunk stream in mid-backward pays it, including c10d's own internal stream when the
    collective is asynchronous).  Worth it only when the payload is large against the step (the unlocked encoder's 440 MB);
  * works unchanged on the gloo backend (CPU tensors) - that is how the N>1 path is tested without GPUs.
"""
import os

import torch
import torch.distributed as dist

UNUSED_PREFIXES = ("get_answer.rnn.", "Bert.bert_model.pooler.")     # never reached by a gradient
EMBED_TABLES = ("fast_embed.weight", "glove_embed.weight")


def init_process_group(device, backend="nccl", **kw):
    """``dist.init_process_group`` for one rank per GPU with RCCL's stream at HIGH priority.  The bucketed all-reduce is on the
    step's critical path (the optimizer waits for it) while the next batch's frozen-encoder GEMMs - 256 workgroups that own every
    CU's registers - arrive on a normal-priority stream; on a normal-priority stream RCCL's kernels queue behind them: measured on
    one MI355X with a world-size-1 group, 23-24 ms per step against 20.2 with the priority raised (19.9 without DP)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        torch.cuda.set_device(torch.device(device))        # every later launch / stream query of this rank targets ITS device
        kw.setdefault("pg_options", dist.ProcessGroupNCCL.Options(is_high_priority_stream=True))
        kw.setdefault("device_id", torch.device(device))
    return dist.init_process_group(backend, **kw)


class GradSync:
    def __init__(self, network, opt, group=None, bucket_bytes=None, pinned_scalar=False, overlap=None):
        """``overlap``: exchange bucket by bucket during backward (hooks + a communication stream) instead of after it; default
        ``opt['dp_overlap_backward']``, off (module docstring).  ``bucket_bytes``: default 16 MB when overlapping, 256 MB otherwise.
        ``pinned_scalar``: opt-in approximation - the caller's optimizer takes ``pinned_sq`` (FusedAdamax.clip_and_step(extra_sq=)) and
        the re-pinned embedding rows are represented by one scalar instead of being exchanged (see the module docstring: exact only
        when the ranks touch disjoint pinned rows).  Default: whole tables exchanged, exact clip norm."""
        self.group = group
        self.world = dist.get_world_size(group)
        self.overlap = bool(opt.get("dp_overlap_backward", False)) if overlap is None else bool(overlap)
        if bucket_bytes is None:
            bucket_bytes = (16 << 20) if self.overlap else (256 << 20)
        self.network = network
        tp = None
        self.mode = "full"
        if "TUNE_PARTIAL" in opt and not opt.get("dp_exchange_pinned_rows"):
            if opt.get("dp_skip_pinned_rows"):
                tp, self.mode = int(opt["tune_partial"]), "zero"
            elif pinned_scalar:
                tp, self.mode = int(opt["tune_partial"]), "scalar"
        entries = []                       # (name, param, rows or None)
        for name, p in network.named_parameters():
            if not p.requires_grad or name.startswith(UNUSED_PREFIXES):
                continue
            rows = tp if (tp is not None and name in EMBED_TABLES) else None
            entries.append((name, p, rows))
        entries.reverse()                  # gradients arrive roughly in reverse registration order
        self.buckets = []
        cur, cur_bytes = [], 0
        for e in entries:
            n = (e[1][:e[2]] if e[2] is not None else e[1]).numel()
            if cur and cur_bytes + 4 * n > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(e)
            cur_bytes += 4 * n
        if cur:
            self.buckets.append(cur)
        # persistent flat buffers and, per parameter, its slice of them (shaped like the exchanged rows)
        self._flat, self._slices = [], []
        for b in self.buckets:
            sizes = [(p[:r] if r is not None else p).numel() for (_, p, r) in b]
            flat = torch.zeros(sum(sizes), dtype=b[0][1].dtype, device=b[0][1].device)
            o, sl = 0, []
            for (_, p, r), n in zip(b, sizes):
                sl.append(flat[o:o + n].view_as(p[:r] if r is not None else p))
                o += n
            self._flat.append(flat)
            self._slices.append(sl)
        self._pinned = [(name, p, r) for b in self.buckets for (name, p, r) in b if r is not None]
        self.pinned_sq = None              # device scalar: sum over ranks of |g[rows >= tune_partial]|^2 / world^2 ("scalar" mode)
        self._sq = None
        self._work = [None] * len(self.buckets)
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for (_, p, _) in b:
                self._bucket_of[p] = bi
                if self.overlap:
                    p.register_post_accumulate_grad_hook(self._make_hook(bi))
        self.payload_bytes = sum(f.numel() * 4 for f in self._flat) + (4 if self.mode == "scalar" else 0)
        self._reset()

    def _reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._streams = [set() for _ in self.buckets]
        self._work = [None] * len(self.buckets)
        self._sq_work = None
        self._next = 0                      # buckets are ALWAYS launched in index order: every rank issues the same
                                            # sequence of collectives even if its gradients become ready in another order

    def _make_hook(self, bi):
        def hook(param):
            if param.is_cuda:               # the stream this gradient was accumulated on (see the module docstring)
                self._streams[bi].add(torch.cuda.current_stream(param.device))
            self._pending[bi] -= 1
            self._launch_ready()
        return hook

    def _launch_ready(self):
        while self._next < len(self.buckets) and self._pending[self._next] <= 0:
            self._launch(self._next)
            self._next += 1

    def _grads(self, bi):
        out = []
        for (_, p, rows) in self.buckets[bi]:
            g = p.grad
            if g is None:                  # a used parameter can still miss a gradient on some batch (e.g. empty group)
                g = torch.zeros_like(p)
                p.grad = g
            out.append(g[:rows] if rows is not None else g)
        return out

    def _comm_stream(self, device):
        """Overlapped mode: the stream the bucket copies and collectives are enqueued from.  NOT one of the trunk's streams - the
        stream a hook happens to run on would have to wait for the other two (``wait_stream`` = everything enqueued there so far,
        far more than this bucket's gradients), serialising the three branches of the backward at every bucket."""
        st = self.__dict__.get("_comm")
        if st is None or st.device != device:
            st = self.__dict__["_comm"] = torch.cuda.Stream(device=device, priority=-1)
        return st

    def _exchange_now(self, bi):
        """Default mode, after backward(): copy, collective and scale of one bucket on the caller's stream (a synchronous c10d
        collective runs on the current stream - no event, no second stream)."""
        flat = self._flat[bi]
        torch._foreach_copy_(self._slices[bi], self._grads(bi))
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        if self.mode == "scalar" and bi == len(self.buckets) - 1:
            parts = [p.grad[rows:].float().pow(2).sum() for (_, p, rows) in self._pinned if p.grad is not None and p.grad.shape[0] > rows]
            self._sq = torch.stack(parts).sum().reshape(1) if parts else torch.zeros(1, device=flat.device)
            dist.all_reduce(self._sq, op=dist.ReduceOp.SUM, group=self.group)

    def _launch(self, bi):
        flat = self._flat[bi]
        grads = self._grads(bi)
        if flat.is_cuda:
            comm = self._comm_stream(flat.device)
            for s in self._streams[bi] | {torch.cuda.current_stream(flat.device)}:
                comm.wait_stream(s)         # everything enqueued on s so far - the accumulations of this bucket included
            for g in grads:
                g.record_stream(comm)
            with torch.cuda.stream(comm):
                torch._foreach_copy_(self._slices[bi], grads)
                self._work[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if self.mode == "scalar" and bi == len(self.buckets) - 1:
                    self._launch_pinned_scalar()
            return
        torch._foreach_copy_(self._slices[bi], grads)
        self._work[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if self.mode == "scalar" and bi == len(self.buckets) - 1:
            self._launch_pinned_scalar()

    def _launch_pinned_scalar(self):
        """One scalar for the rows that are never exchanged: this rank's sum of squares of their gradients."""
        dev = self._flat[0].device
        if dev.type == "cuda":
            # the tables' gradients may have been accumulated on other trunk streams than the one this (last) bucket's hook runs
            # on, and in EARLIER buckets: wait for every stream any bucket holding a pinned table noted
            cur = torch.cuda.current_stream(dev)
            for bi, b in enumerate(self.buckets):
                if any(r is not None for (_, _, r) in b):
                    for s in self._streams[bi]:
                        if s != cur:
                            cur.wait_stream(s)
        parts = []
        for (_, p, rows) in self._pinned:
            g = p.grad
            if g is not None and g.shape[0] > rows:
                parts.append(g[rows:].float().pow(2).sum())
        sq = torch.stack(parts).sum().reshape(1) if parts else torch.zeros(1, device=dev)
        self._sq = sq
        self._sq_work = dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def average_gradients(self):
        """Call after ``loss.backward()``: finishes the outstanding bucket all-reduces and writes the averaged
        gradients back in place."""
        while self.overlap and self._next < len(self.buckets):   # gradients that never fired a hook (unused on this batch): still in order
            self._launch(self._next)
            self._next += 1
        inv = 1.0 / self.world
        for bi in range(len(self.buckets)):
            if self.overlap:
                self._work[bi].wait()
            else:
                self._exchange_now(bi)
            flat = self._flat[bi]
            if self.world > 1:
                flat.mul_(inv)                         # one launch per bucket
            for sl, (_, p, rows) in zip(self._slices[bi], self.buckets[bi]):
                if rows is None:
                    p.grad = sl                        # the averaged gradient IS the bucket slice: no copy back
                else:
                    p.grad[:rows].copy_(sl)
                    if self.mode == "zero":
                        p.grad[rows:].zero_()
        if self._sq_work is not None or (self._sq is not None and not self.overlap):
            if self._sq_work is not None:
                self._sq_work.wait()
            self.pinned_sq = self._sq.mul_(inv * inv) if self.world > 1 else self._sq
        self._reset()

    def broadcast_parameters(self, src=0):
        """Make every replica start from rank ``src``'s weights (and buffers)."""
        for t in list(self.network.parameters()) + list(self.network.buffers()):
            dist.broadcast(t.data, src=src, group=self.group)
