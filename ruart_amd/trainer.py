"""``SDNetTrainer`` / ``BaseTrainer`` - the reference's training-loop surface (Models/SDNetTrainer.py:29-518,
Models/BaseTrainer.py:6-69) over the MI355X hot path, plus single-node data parallelism.

Kept from the reference: ``setup_model(vocab_embedding)``, ``update(batch, batch_i) -> float``, ``predict(batch)``,
``evaluate(data_loader, ...)``, ``ToCUDA(batch)``, ``load_model(path)``, ``save_for_predict(path)``,
``instance_bce_with_logits`` (the probabilities are fed to BCE-with-logits and scaled by the label width, :510-518),
optimizer selection by ``opt['optimizer']`` (:307-317), global-norm clipping (:366), re-pinning of embedding rows
>= ``tune_partial`` after every step (:369-373), checkpoint format (:492-509, 453-466).

New: ``world_size > 1`` => one process per GPU, gradients averaged with RCCL (dp.GradSync) before clipping; every rank
holds identical parameters, so the step is "N independent B-sample reference steps with averaged gradients"
(SURVEY.md section 8e).  ``train()`` / ``predict_for_test()`` without arguments follow the reference end to end - vocabulary and
word vectors from ``train_meta.msgpack``, records from ``{train,val,test}-preprocessed.msgpack`` through ``dataset.VQA_Dataset``,
run folder, best-model checkpoints, ``submission.json`` - and ``train`` also takes any iterable of collated batches.  The offline
preprocessing itself (spaCy / fastText / detector outputs -> msgpack) stays out of scope.
"""
import json
import logging
import os
import random
import time

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from . import layers as L
from .batch import VQA_collate, to_device
from .sampler import VQA_Sampler
from .sdnet import SDNet

log = logging.getLogger(__name__)


class _PendingLoss:
    """What ``update`` returns while the readback of its step is deferred: ``float()`` / ``item()`` resolve it (a device sync if the
    step is still running)."""

    def __init__(self, trainer, pending):
        self._trainer, self._pending = trainer, pending

    def item(self):
        if self._pending["value"] is None:          # still the trainer's outstanding step: resolve it (and only it) now
            self._trainer.flush_readback()
        return self._pending["value"]

    __float__ = item

    def __format__(self, spec):
        return format(self.item(), spec)

    def __repr__(self):
        return repr(self.item())

    # arithmetic and comparisons resolve the value (the reference's update() returns a float)
    def __add__(self, o): return self.item() + o
    def __radd__(self, o): return o + self.item()
    def __sub__(self, o): return self.item() - o
    def __rsub__(self, o): return o - self.item()
    def __mul__(self, o): return self.item() * o
    def __rmul__(self, o): return o * self.item()
    def __truediv__(self, o): return self.item() / o
    def __rtruediv__(self, o): return o / self.item()
    def __neg__(self): return -self.item()
    # comparisons with anything that is not a number answer as a plain float's would (NotImplemented -> False / TypeError from
    # Python itself), and the value hashes as the float it stands for: ``loss == None``, ``loss in some_list`` and set / dict
    # membership stay legal, as they are for the float the reference's update() returns
    @staticmethod
    def _num(o):
        if isinstance(o, _PendingLoss):
            return o.item()
        if isinstance(o, (bool, int, float)):
            return float(o)
        try:
            import numbers
            if isinstance(o, numbers.Real) or (hasattr(o, "__float__") and getattr(o, "ndim", 0) == 0):
                return float(o)
        except (TypeError, ValueError):
            pass
        return None

    def _cmp(self, o, op):
        v = self._num(o)
        return NotImplemented if v is None else op(self.item(), v)

    def __lt__(self, o): return self._cmp(o, lambda a, b: a < b)
    def __le__(self, o): return self._cmp(o, lambda a, b: a <= b)
    def __gt__(self, o): return self._cmp(o, lambda a, b: a > b)
    def __ge__(self, o): return self._cmp(o, lambda a, b: a >= b)
    def __eq__(self, o): return self._cmp(o, lambda a, b: a == b)
    def __hash__(self): return hash(self.item())


class AverageMeter:
    """Utils/CoQAUtils.py:837-858."""

    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def decode_predictions(scores, num_cnt, extra_info, opt):
    """The answer decode of Models/SDNetTrainer.py:391-450 for scores (B, n_slots) on any device: per sample the reference walks
    the slots by descending score, skips the ``<OCR>`` sentinel and the padding slots (>= num_cnt), stops at the first real OCR
    token - or at the no-answer slot (last column, ``label_no_answer``) if that comes first.  A descending walk that stops at the
    first admissible slot is a masked arg-max, done here in one device op for the batch.  When nothing is admissible the
    reference's loop runs out and is left with the LOWEST-scored slot; that case is reproduced too.
    Returns (ANLS sum, ACC sum, res, save_res) as the reference's ``predict`` does after the loss."""
    from .metrics import note_stvqa, note_textvqa
    if "label_yesno" in opt or "fixed_answers" in opt:
        raise NotImplementedError("yes/no and fixed-answer slots are outside the accelerated path (SURVEY section 8)")
    B, n_slots = scores.shape
    no_answer = "label_no_answer" in opt
    cnt = torch.as_tensor(list(num_cnt), device=scores.device)
    sentinel = torch.as_tensor([len(e["ocr_list"]) - 1 for e in extra_info], device=scores.device)
    ar = torch.arange(n_slots, device=scores.device).unsqueeze(0)
    valid = (ar < cnt.unsqueeze(1)) & (ar != sentinel.unsqueeze(1))
    if no_answer:
        valid[:, -1] = True
    best = scores.masked_fill(~valid, -1.0).argmax(dim=1)
    idxs = torch.where(valid.any(dim=1), best, scores.argmin(dim=1)).cpu().tolist()
    prob = scores.detach().cpu()
    res, save_res, ANLS, ACC = [], [], 0, 0
    for i, idx in enumerate(idxs):
        answer = extra_info[i]["ocr_list"][idx] if idx < int(num_cnt[i]) else "unanswerable"
        res.append({"question_id": extra_info[i]["q_id"], "answer": answer})
        save_res.append({"question_id": extra_info[i]["q_id"], "prediction": answer, "answers": extra_info[i]["answers"],
                         "score": prob[i, idx].item(), "idx": idx, "ids_len": n_slots, "ocr_list": extra_info[i]["ocr_list"]})
        if extra_info[i]["answers"] is not None:
            a = note_stvqa(extra_info[i]["answers"], answer)
            c = note_textvqa(extra_info[i]["answers"], answer)
            ACC += min(c * 10 / 3.0, 1) if len(extra_info[i]["answers"]) == 10 else min(c * 10, 1)
            ANLS += a if a >= 0.5 else 0
    return ANLS, ACC, res, save_res


class BaseTrainer:
    """Models/BaseTrainer.py:6-69: option plumbing, the feature folder, the run folder."""

    def __init__(self, opt):
        self.opt = opt
        self.isTrain = False
        self.use_cuda = opt.get("cuda") is True
        self.saveFolder = opt.get("saveFolder", ".")
        self.opt.setdefault("logFile", "log.txt")
        if "source_dir" in opt and "datadir" in opt:     # :20-21; a caller-provided FEATURE_FOLDER stands otherwise
            opt["FEATURE_FOLDER"] = os.path.join(opt["datadir"], "./source/data/" + opt["source_dir"] + "/")

    def getSaveFolder(self):
        """Training: the first free ``<datadir>/conf~/run_<n>``; otherwise the folder the loaded model lives in (:49-62)."""
        if self.isTrain:
            runid = 1
            while os.path.exists(os.path.join(self.opt["datadir"], "conf~", "run_" + str(runid))):
                runid += 1
            self.saveFolder = os.path.join(self.opt["datadir"], "conf~", "run_" + str(runid))
            os.makedirs(self.saveFolder)
            log.info("Saving logs, model and evaluation in %s", self.saveFolder)
        else:
            self.saveFolder = os.path.join(self.opt["datadir"], "/".join(self.opt["MODEL_PATH"].split("/")[:2]))
            os.makedirs(self.saveFolder, exist_ok=True)

    def saveConf(self):
        if "confFile" not in self.opt:
            return
        with open(self.opt["confFile"], encoding="utf-8") as f, \
                open(os.path.join(self.saveFolder, "conf_copy"), "w", encoding="utf-8") as fw:
            fw.writelines(f)


class SDNetTrainer(BaseTrainer):
    def __init__(self, opt, device=None, process_group=None):
        super().__init__(opt)
        L.set_dropout_prob(0.0 if "DROPOUT" not in opt else float(opt["DROPOUT"]))
        self.seed = int(opt["SEED"])
        random.seed(self.seed)
        np.random.seed(self.seed)
        torch.manual_seed(self.seed)
        self._rank_seeded = False
        self.batch_size = opt["batch_size"]
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type == "cuda":
            # one process drives ONE GPU: make it the current device, so that every stream query, allocation and kernel launch
            # below (the ctypes-launched HIP kernels take torch's current stream) targets this rank's device
            torch.cuda.set_device(self.device if self.device.index is not None else torch.cuda.current_device())
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.process_group = process_group
        self.grad_sync = None
        self.fixed_answers_len = 0
        self.updates = 0
        self.best_ANLS = self.best_ACC = -1
        self.best_ANLS_batch = self.best_ACC_batch = -1

    # -- model / optimizer ------------------------------------------------------------------------------------
    def setup_model(self, vocab_embedding):
        self.train_loss = AverageMeter()
        self.staged = None                 # what update(stage_next=...) staged last
        opt = dict(self.opt)
        opt["device"] = self.device
        self.network = SDNet(opt, vocab_embedding).to(self.device)
        for name in ("fixed_embedding_fast", "fixed_embedding_glove"):
            if hasattr(self.network, name):
                setattr(self.network, name, getattr(self.network, name).to(self.device))
        params = [p for p in self.network.parameters() if p.requires_grad]
        o = self.opt["optimizer"]
        if o == "ADAM":
            self.optimizer = optim.Adamax(params, weight_decay=0.5, lr=1e-3)
        elif o == "#":
            if self.device.type == "cuda" and self.opt.get("ruart_fused_optimizer", True):
                from .optim import FusedAdamax
                pinned = {}
                if "TUNE_PARTIAL" in self.opt:          # rows >= tune_partial are re-pinned after every step: never updated
                    for flag, name in (("FastText", "fast_embed"), ("GLOVE", "glove_embed")):
                        if flag in self.opt and getattr(self.network, name).weight.requires_grad:
                            pinned[getattr(self.network, name).weight] = self.opt["tune_partial"]
                self.optimizer = FusedAdamax(params, lr=self.opt.get("lr", 2e-3), pinned=pinned)
            else:
                self.optimizer = optim.Adamax(params, lr=self.opt.get("lr", 2e-3))
        elif o == "ADAM2":
            self.optimizer = optim.Adam(params, lr=self.opt.get("lr", 1e-3))
        elif o == "SGD":
            self.optimizer = optim.SGD(params, lr=self.opt["lr"])
        else:
            raise ValueError("optimizer is wrong")
        if self.opt["loss"] in ("BCE", "BCE_D1"):
            self.loss_func = self.instance_bce_with_logits
        elif self.opt["loss"] == "CE":
            self.loss_func = nn.functional.cross_entropy
        else:
            raise ValueError("loss parameter is error")
        self.updates = 0
        # opt['ruart_dp'] = False: a plain single-replica trainer even inside an initialised process group (tests compare against it)
        if self.opt.get("ruart_dp", True) and (self.process_group is not None or (
                torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1)):
            from .dp import GradSync
            # default: the embedding tables are exchanged whole, so the clip norm is the exact norm of the averaged gradient;
            # opt['dp_pinned_scalar'] (fused optimizer only) represents the re-pinned rows by one scalar instead (dp.py)
            self.grad_sync = GradSync(self.network, self.opt, group=self.process_group,
                                      pinned_scalar=bool(self.opt.get("dp_pinned_scalar")) and hasattr(self.optimizer, "clip_and_step"))
            self.grad_sync.broadcast_parameters()
            # the replicas share their parameters, not their dropout masks: every rank seeded torch identically above (so that the
            # initial weights agree even before the broadcast), which would make all ranks draw the SAME variational-dropout masks
            # step after step.  From here on each rank's generators run their own stream.
            rank = torch.distributed.get_rank(self.process_group)
            if rank > 0 and not self._rank_seeded:
                torch.manual_seed(self.seed + 7919 * rank)
            self._rank_seeded = True

    def close(self, final=False):
        """End of a session: the deferred loss check, the encoder pass still running ahead (``Bert.close``).  ``final=True`` - the
        END OF THE PROCESS's use of this trainer (bench.py, the tools, ``__graft_entry__``, a script's last call) - also destroys the
        CU-masked run-ahead stream (hipExtStreamCreateWithCUMask, the default in the fp16c mode): alive at static destruction it takes a
        process with a tool library loaded (rocprofv3, any other HSA tool) down in __cxa_finalize.  Between
        sessions (``train()``, a stand-alone ``evaluate()``, ``predict_for_test()`` call this without ``final`` on every exit path) the
        stream is KEPT: a re-created one can land on a trunk stream's hardware queue slot (``Bert.close``).  The decision is the
        caller's, not a guess from the environment (round 5 looked for "rocprof" in environment variables; RUART_DESTROY_STREAMS=1
        still forces the destruction for experiments)."""
        import sys
        unwinding = sys.exc_info()[0] is not None      # called from a ``finally`` while another exception propagates
        try:
            self.flush_readback()
        except AssertionError as e:
            if not unwinding:
                raise                        # nothing else is propagating: the deferred NaN / loss assert IS the error
            print("ruart_amd: deferred loss check failed during unwinding: %s" % (e,), file=sys.stderr)
        finally:
            bert = getattr(getattr(self, "network", None), "Bert", None)
            if bert is not None:
                bert.close(destroy=True if final else None)
            if final:
                self._destroy_masked_streams()

    def _destroy_masked_streams(self):
        """``close(final=True)``: the step's CU-masked streams (SDNet.trunk_stream_cus) go the way of the encoder's masked stream."""
        from . import hip
        for cache in (self.__dict__.get("_step_streams"), getattr(getattr(self, "network", None), "__dict__", {}).get("_streams")):
            for key, v in list((cache or {}).items()):
                sts = v if isinstance(v, tuple) else (v,)
                if any(getattr(s, "_ruart_masked", False) for s in sts):
                    del cache[key]
                    for s in sts:
                        hip.destroy_stream(s)

    def ToCUDA(self, batch):
        """Models/SDNetTrainer.py:208-230.  Index vectors and the packed BERT stream are prepared here, from the host copies,
        before the tensors move - unless the batch already carries them (``VQA_collate(opt, prepare_index=True)`` builds the
        host part in the DataLoader workers; then this is copies only)."""
        q, ocr, od = batch[0], batch[1], batch[2]
        if hasattr(self, "network"):
            self.network.prepare(q, ocr, od)
        out = to_device(batch, self.device)
        if "_ruart_index" in q:
            out[0]["_ruart_index"] = q["_ruart_index"]
        return out

    def instance_bce_with_logits(self, logits, labels):
        assert logits.dim() == 2
        loss = nn.functional.binary_cross_entropy_with_logits(logits, labels)
        if self.opt["loss"] == "BCE_D1":
            loss = loss * labels.size(1)
        return loss

    # -- one optimizer step -----------------------------------------------------------------------------------
    def update(self, batch, batch_i=0, next_batch=None, stage_next=None):
        """One optimizer step.  ``stage_next`` (optional callable): run once after the whole step has been enqueued and BEFORE the
        host waits for its loss - the place to fetch the batch after next from the loader and ship it (``ToCUDA``): that costs
        0.5-0.7 ms of host time per step, which delays the next encoder launch by as much when it sits between two ``update`` calls and
        nothing when it sits where the host waits anyway.  Its return value is kept in ``self.staged``.  ``next_batch`` (already through ToCUDA) lets the frozen BERT pass of the following step run
        concurrently with this step's SDNet trunk, on its own stream.  The step runs on a stream of its own too, whose priority ``SDNet.trunk_stream_priority``
        chooses (see there and DESIGN.md section 5)."""
        self.network.train()                      # (the step stream's priority depends on the mode: set it before choosing)
        return self.on_step_stream(self._update, batch, batch_i, next_batch, stage_next)

    def on_step_stream(self, fn, *args):
        """Run ``fn(*args)`` - a training step or an evaluation forward - with the trainer's step stream current, and join it with the
        caller's stream on both sides.  Needed for the encoder pass that runs one batch ahead to overlap anything: its CU-masked stream
        (hipExtStreamCreateWithCUMask takes no flags) is a BLOCKING stream, i.e. every launch on the legacy default stream - torch's
        current stream unless told otherwise - waits for all of the encoder's enqueued work and the two run strictly one after the
        other (forward-only steps: 27 ms = 20 + 7, `tools/step_timeline.py --fwd`).  The step stream is a non-blocking pool stream.
        (A trainable encoder: nothing runs ahead, the caller's stream is used as it is.)"""
        st = self._step_stream()
        if st is None:
            return fn(*args)
        cur = torch.cuda.current_stream(self.device)
        if cur == st:                       # the caller already runs on the step stream (train(), step_stream()): nothing to join
            return fn(*args)
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            out = fn(*args)
        cur.wait_stream(st)
        return out

    def _step_stream(self):
        """The step stream of the network's CURRENT mode (created on first use; one per priority: the process keeps the streams it
        starts with, profiles/HISTORY.md round 5 (6b) / (9)); None on the CPU and with a trainable encoder (nothing runs ahead there)."""
        dev = self.device
        unlocked = getattr(getattr(self.network, "Bert", None), "bert_model", None) is not None
        if dev.type != "cuda" or unlocked:
            return None
        pr = self.network.trunk_stream_priority()       # (differs between training and evaluation in the fp16c schedule)
        cache = self.__dict__.setdefault("_step_streams", {})
        st = cache.get(pr)
        if st is None:
            ncu = self.network.trunk_stream_cus()                   # an experiment's CU mask (SDNet.trunk_stream_cus; 0 by default)
            if ncu != 0:
                from . import hip
                st = hip.cu_masked_stream(ncu, dev)
            elif pr > 0:                                        # LOW priority: not in torch's pool
                from . import hip
                st = hip.priority_stream(pr, dev)
            else:
                st = torch.cuda.Stream(device=dev, priority=pr)
            cache[pr] = st
        return st

    def step_stream(self):
        """Context manager: the TRAINING step stream as torch's current stream - for a loop that stages batches and calls ``update``
        many times (``train()`` runs inside it; a script that drives ``update`` itself should too).  Called from torch's default
        stream, every ``update`` has to join that stream with the step stream on both sides, and the default stream is the LEGACY
        stream: each of those two markers waits for - and holds back - the CU-masked encoder stream (a blocking stream), so the
        next step's trunk cannot start before the encoder pass that runs beside it has ended.  With the caller on the step stream
        the first kernels of step t+1 (embeddings, the question branch) run under the tail of that pass: 22.75 -> 22.00 ms per step
        (interleaved, three rounds; profiles/HISTORY.md round 5 (12)).  Everything the loop enqueues - ``ToCUDA`` copies included - is then
        ordered on that one stream; results handed to another stream need the usual ``wait_stream``."""
        import contextlib
        self.network.train()
        st = self._step_stream()
        return torch.cuda.stream(st) if st is not None else contextlib.nullcontext()

    def _update(self, batch, batch_i, next_batch, stage_next=None):
        self.network.train()
        self.network.drop_emb = True
        q_list, ocr_list, od_list, targets, extra_info = batch
        if next_batch is not None:
            self.network.prefetch_bert(next_batch[0], next_batch[1], next_batch[2])
        scores, _ = self.network(q_list, ocr_list, od_list)
        if self.opt["loss"] == "CE":
            targets = torch.nonzero(targets)[:, 1]
        loss = self.loss_func(scores, targets)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync.average_gradients()
        if hasattr(self.optimizer, "clip_and_step"):       # fused: global-norm clip + Adamax in three launches
            self.optimizer.clip_and_step(self.opt["grad_clipping"],
                                         extra_sq=self.grad_sync.pinned_sq if self.grad_sync is not None else None)
        else:
            torch.nn.utils.clip_grad_norm_(self.network.parameters(), self.opt["grad_clipping"])
            self.optimizer.step()
        self.updates += 1
        if "TUNE_PARTIAL" in self.opt:
            # Models/SDNetTrainer.py:369-373 re-pins the rows >= tune_partial after every step.  The fused optimizer never writes them
            # (FusedAdamax(pinned=): their update is left out), so they still hold the pinned values: the two 23 MB copies per step
            # are skipped for the tables it knows as pinned
            tp = self.opt["tune_partial"]
            kept = getattr(self.optimizer, "pinned", {})
            if "FastText" in self.opt and id(self.network.fast_embed.weight) not in kept:
                self.network.fast_embed.weight.data[tp:] = self.network.fixed_embedding_fast
            if "GLOVE" in self.opt and id(self.network.glove_embed.weight) not in kept:
                self.network.glove_embed.weight.data[tp:] = self.network.fixed_embedding_glove
        if self._defer_readback():
            lazy = self._readback_later(loss, stage_next)
            return lazy
        if stage_next is not None:
            self.staged = stage_next()
        self.host_enqueued_at = time.perf_counter()      # (bench.py: how long the host took to enqueue the step)
        # the reference's NaN contract (SDNetTrainer.py:339-359 + the asserts inside forward): one sync, here
        loss_val = loss.item()
        self.network.check_nan()
        assert loss_val == loss_val, "loss nan"
        self.train_loss.update(loss_val, 1)
        return loss_val

    # -- loss / NaN readback one step late (trained encoder) ------------------------------------------------------------
    # With the encoder trained nothing runs ahead of a step: encoder forward -> trunk -> encoder backward is one chain, its two
    # encoder parts bound by the GPU (10 + 21 ms of GEMMs, the host idle) and its trunk part by the host (~15 ms of enqueueing
    # ~850 small launches, the GPU mostly idle).  A ``loss.item()`` at the end of every step welds the two together; reading the
    # loss and the NaN flag of step t at the END of step t+1 instead lets the host enqueue the trunk of step t+1 while the GPU is
    # still in the encoder backward of step t.  The reference's contract is kept one step late: the same asserts fire, before
    # any checkpoint or evaluation can see the weights (``flush_readback`` runs first there), and ``train_loss`` sees every
    # step's value in order.  The frozen-encoder pipeline keeps its per-step sync: there it was measured FASTER.
    # Round 5: since the training loop runs on the step stream (step_stream) the frozen pipeline's trunk chain is as long as its encoder
    # chain, and the host sits on it once per step (loss.item(): 22.28 ms per step against 22.03 deferred, and every host hiccup is a
    # slow step).  ``train()`` therefore defers inside its own loop - it only logs the loss, and flush_readback keeps the asserts ahead
    # of every evaluation and checkpoint; a direct ``update()`` call keeps the reference's contract (a float, asserts in the same
    # step) unless opt['ruart_defer_readback'] says otherwise.
    def _defer_readback(self):
        d = self.opt.get("ruart_defer_readback")
        if d is None:
            d = getattr(getattr(self.network, "Bert", None), "bert_model", None) is not None or getattr(self, "_in_train_loop", False)
        return bool(d) and self.device.type == "cuda"

    def _readback_later(self, loss, stage_next=None):
        from . import ops
        slots = self.__dict__.setdefault("_rb_slots", [torch.empty(2, dtype=torch.float32).pin_memory() for _ in range(2)])
        host = slots[self.updates & 1]
        flag = ops.nan_flag.flag
        host[0:1].copy_(loss.detach().reshape(1), non_blocking=True)
        if flag is not None and flag.device == loss.device:
            host[1:2].copy_(flag, non_blocking=True)        # int32 -> float32: 0 stays 0
            flag.zero_()
        else:
            host[1] = 0.0
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        prev, self._rb_pending = self.__dict__.get("_rb_pending"), {"host": host, "event": ev, "value": None}
        lazy = _PendingLoss(self, self._rb_pending)
        if stage_next is not None:
            self.staged = stage_next()
        self.host_enqueued_at = time.perf_counter()      # (bench.py: how long the host took to enqueue the step)
        if prev is not None:
            self._resolve(prev)
        return lazy

    def _resolve(self, pending):
        if pending["value"] is None:
            pending["event"].synchronize()
            host = pending["host"]              # (reused two steps later: the value is taken out now)
            loss_val, bad = float(host[0]), float(host[1])
            assert bad == 0, "NaN produced inside the SDNet kernels (reference: assert torch.sum(torch.isnan(x)) == 0)"
            assert loss_val == loss_val, "loss nan"
            self.train_loss.update(loss_val, 1)
            pending["value"] = loss_val
        return pending["value"]

    def flush_readback(self):
        """Resolve the step whose loss / NaN flag have not been read back yet (no-op otherwise); returns its loss or None."""
        pending, self._rb_pending = self.__dict__.get("_rb_pending"), None
        return self._resolve(pending) if pending is not None else None

    # -- inference --------------------------------------------------------------------------------------------
    def predict(self, batch, all_ans=False, next_batch=None):
        """Models/SDNetTrainer.py:378-451: arg-max over VALID answer slots, ANLS / ACC when answers are known.
        ``next_batch`` (already through ToCUDA): its frozen-encoder pass is started beside this batch's trunk, as in ``update``."""
        self.network.eval()
        return self.on_step_stream(self._predict, batch, all_ans, next_batch)

    def _predict(self, batch, all_ans=False, next_batch=None):
        self.flush_readback()                  # (a deferred training step's asserts come before anything reads the weights)
        self.network.eval()
        self.network.drop_emb = False
        q_list, ocr_list, od_list, gt_list, extra_info = batch
        if next_batch is not None:
            self.network.prefetch_bert(next_batch[0], next_batch[1], next_batch[2])
        with torch.no_grad():
            scores, _ = self.network(q_list, ocr_list, od_list)
            loss = self.loss_func(scores, gt_list).item() if gt_list is not None else 0
        self.network.check_nan()
        ANLS, ACC, res, save_res = decode_predictions(scores, ocr_list["num_cnt"], extra_info, self.opt)
        return loss, ANLS, ACC, res, save_res

    def _is_main(self):
        """Rank 0 (or no process group): the only rank that creates the run folder and writes checkpoints / predictions."""
        d = torch.distributed
        return not (d.is_available() and d.is_initialized()) or d.get_rank(self.process_group) == 0

    def _loader(self, data, sampler, workers=0):
        from torch.utils.data import DataLoader
        collate = VQA_collate(self.opt, prepare_index=workers > 0).VQA_collate_fun
        # pinned batches (tensors and, through BatchIndex.pin_memory, the host-built index): ToCUDA becomes asynchronous copies
        return DataLoader(data, batch_sampler=sampler, collate_fn=collate, num_workers=workers,
                          pin_memory=self.device.type == "cuda")

    def evaluate(self, val_data, batch_i=0, mode="dev"):
        """Models/SDNetTrainer.py:127-176.  ``val_data`` is a ``VQA_Dataset`` (batched here in the reference's deterministic
        order; ``dev`` keeps the best-ANLS / best-ACC checkpoints and ``save_res_last.json``, ``test`` writes
        ``submission.json`` without the wrap-around padding of the last batch) or any iterable of collated batches (metrics
        only).  Returns (mean loss, ANLS, ACC, predictions)."""
        assert mode in ("train", "dev", "test")
        try:
            return self._evaluate(val_data, batch_i, mode)
        finally:
            if not getattr(self, "_in_train", False):     # stand-alone evaluation: nothing else will release the stream
                self.close()
            elif self.device.type == "cuda" and self.opt.get("ruart_empty_cache_after_eval", True):
                # an evaluation inside train(): its no-grad forwards leave the caching allocator with a block layout the training steps
                # would go on reusing - measured 0.7-1.0 ms per training step under the schedule of the time (bench.py's parity check,
                # profiles/HISTORY.md round 5 (11); under the final schedule of (12) the effect was no longer measurable); handing the cached
                # blocks back costs a few re-allocations in the next step
                torch.cuda.synchronize(self.device)
                torch.cuda.empty_cache()

    def _evaluate(self, val_data, batch_i, mode):
        from torch.utils.data import Dataset
        is_dataset = isinstance(val_data, Dataset)
        loader = self._loader(val_data, VQA_Sampler(val_data, self.opt["max_batch_num"], self.batch_size, False)) if is_dataset \
            else val_data
        loss = ANLS = ACC = n = nb = 0
        res, save_res = [], []
        it = iter(loader)
        nxt = next(it, None)
        nxt = self.ToCUDA(nxt) if nxt is not None else None
        while nxt is not None:
            batch = nxt
            nxt = next(it, None)
            nxt = self.ToCUDA(nxt) if nxt is not None else None       # lookahead: the next batch's encoder pass runs beside this trunk
            l, a, c, r, sr = self.predict(batch, next_batch=nxt)
            loss, ANLS, ACC, n, nb = loss + l, ANLS + a, ACC + c, n + len(r), nb + 1
            res.extend(r)
            save_res.extend(sr)
        n_items = len(val_data) if is_dataset else n
        loss, ANLS, ACC = loss / max(nb, 1), ANLS / max(n_items, 1), ACC / max(n_items, 1)
        if not is_dataset or not self._is_main():
            return loss, ANLS, ACC, res
        if mode == "test":
            end = len(val_data) % self.batch_size
            if end != 0:
                res = res[:-(self.batch_size - end)]
            path = os.path.join(self.saveFolder, "submission.json")
            with open(path, "w") as wf:
                json.dump(res, wf, indent=2)
            log.info("%d test samples are predicted, %d predictions saved in %s", len(val_data), len(res), path)
            return loss, ANLS, ACC, res
        if mode == "dev":
            with open(os.path.join(self.saveFolder, "save_res_last.json"), "w") as wf:
                json.dump(save_res, wf, indent=2)
            if ANLS > self.best_ANLS:
                self.best_ANLS, self.best_ANLS_batch = ANLS, batch_i
                self.save_for_predict(os.path.join(self.saveFolder, "ANLS_best_model.pt"))
            if ACC > self.best_ACC:
                self.best_ACC, self.best_ACC_batch = ACC, batch_i
                self.save_for_predict(os.path.join(self.saveFolder, "ACC_best_model.pt"))
        log.info("Dataset: %s Batch: %7d ANLS: %.3f Best ANLS: %.3f Batch: %d ACC: %.3f Best ACC:%.3f Batch:%d", mode, batch_i, ANLS,
                 self.best_ANLS, self.best_ANLS_batch, ACC, self.best_ACC, self.best_ACC_batch)
        return loss, ANLS, ACC, res

    def _setup_from_meta(self):
        from .dataset import load_meta
        self.vocab, self.char_vocab, vocab_embedding = load_meta(self.opt)
        self.setup_model(vocab_embedding)

    def _records(self, split):
        from .dataset import load_msgpack
        return load_msgpack(os.path.join(self.opt["FEATURE_FOLDER"], split + "-preprocessed.msgpack"))["data"]

    def train(self, train_loader=None, val_loader=None, eval_every=1500, log_every=30):
        """The outer loop of Models/SDNetTrainer.py:50-123.  Without arguments it is the reference's ``train()``: run folder,
        model from ``train_meta.msgpack`` (+ ``RESUME``), ``VQA_Dataset`` over the train / val msgpack records, the deterministic
        ``VQA_Sampler`` stream, evaluation every ``eval_every`` batches and once more on both sets at the end.  With a loader it
        runs over any iterable of collated batches.  Either way the next batch is staged one step ahead so that its frozen-BERT
        pass overlaps this step's trunk."""
        self.isTrain = True
        self._in_train = True
        try:
            self._train(train_loader, val_loader, eval_every, log_every)
        finally:
            self._in_train = False
            self._in_train_loop = False
            if self.__dict__.pop("_gc_frozen", False):
                import gc
                gc.unfreeze()                # the permanent generation of _train is handed back: train() leaves the interpreter as it found it
            self.close()

    def _train(self, train_loader, val_loader, eval_every, log_every):
        train_data = None
        batch_st = 0
        if train_loader is None:
            from .dataset import VQA_Dataset
            if self._is_main():
                self.getSaveFolder()
                self.saveConf()
            self._setup_from_meta()
            if "RESUME" in self.opt:
                self.load_model(os.path.join(self.opt["datadir"], self.opt["MODEL_PATH"]))
            batch_st = self.opt.get("batch_st", 0)
            train_data = VQA_Dataset(self._records("train"), self.opt)
            dist = torch.distributed
            rank, world = (dist.get_rank(self.process_group), dist.get_world_size(self.process_group)) \
                if dist.is_available() and dist.is_initialized() else (0, 1)
            sampler = VQA_Sampler(train_data, self.opt["max_batch_num"], self.batch_size, True, batch_st=batch_st,
                                  epoch=self.opt.get("epoch"), rank=rank, world_size=world)
            train_loader = self._loader(train_data, sampler, workers=self.opt.get("num_worker", 0))
            val_loader = VQA_Dataset(self._records("val"), self.opt)
        it = iter(train_loader)
        # Everything alive at this point - the model, the optimizer state, the dataset records with the reference's nested python lists -
        # stays alive for the whole run: move it out of the garbage collector's sight (a permanent generation), or every full collection
        # walks all of it and stops the host for ~45 ms, i.e. one 70 ms step in ~25 (measured).  The
        # collector keeps running over what the steps themselves create.  opt['ruart_gc_freeze'] = False leaves the interpreter alone.
        if self.opt.get("ruart_gc_freeze", True):
            import gc
            gc.collect()
            gc.freeze()
            self._gc_frozen = True           # (train()'s finally unfreezes)

        def stage():                                                  # the next batch of the loader, shipped to the device (or None)
            b = next(it, None)
            return self.ToCUDA(b) if b is not None else None

        # the whole loop - staging copies, steps, the evaluations in between - with the training step stream current: no step has to be
        # joined with torch's (legacy) default stream, whose markers would hold the next step back until the encoder pass beside it
        # has ended (step_stream)
        with self.step_stream():
            self._in_train_loop = True                                    # (the loss of a step is read back one step late: _defer_readback)
            batch, nxt = stage(), None
            if batch is not None:
                nxt = stage()                                             # one batch of lookahead feeds the BERT prefetch
            batch_i = batch_st
            while batch is not None:
                if val_loader is not None and batch_i % eval_every == 0:
                    self.evaluate(val_loader, batch_i)
                # the batch after next is fetched and shipped inside update(), where the host waits for the step anyway
                loss = self.update(batch, batch_i, next_batch=nxt, stage_next=stage if nxt is not None else None)
                batch, nxt = nxt, (self.staged if nxt is not None else None)
                self.staged = None
                if batch_i % log_every == 0:
                    # (a deferred loss is resolved by float(): the running average beside it then includes this step, as in the reference)
                    cur = float(loss)
                    log.info("updates[%6d] train loss[%8.5f / %8.5f]", self.updates, self.train_loss.avg, cur)
                batch_i += 1
            self._in_train_loop = False
            self.flush_readback()                            # the last step's loss and asserts
            if train_data is not None:                       # :121-122
                self.evaluate(val_loader, batch_i - 1)
                self.evaluate(train_data, batch_i - 1, mode="train")
                log.info("Training over")

    def predict_for_test(self):
        """Models/SDNetTrainer.py:231-251: load the model named by ``MODEL_PATH``, predict the test records, write
        ``submission.json`` next to it."""
        from .dataset import VQA_Dataset
        assert "RESUME" in self.opt
        self.isTrain = False
        self.getSaveFolder()
        self._setup_from_meta()
        self.load_model(os.path.join(self.opt["datadir"], self.opt["MODEL_PATH"]))
        test_data = VQA_Dataset(self._records("test"), self.opt, mode="test")
        return self.evaluate(test_data, 0, "test")          # a stand-alone evaluate(): closes the run-ahead stream on exit

    # -- checkpoints ------------------------------------------------------------------------------------------
    def load_model(self, model_path):
        """:453-466 - tolerant load: unknown keys dropped, missing keys keep their current value."""
        ckpt = torch.load(model_path, map_location="cpu")
        state = ckpt["state_dict"]["network"]
        cur = self.network.state_dict()
        state = {k: v for k, v in state.items() if k in cur}
        for k, v in cur.items():
            state.setdefault(k, v)
        self.network.load_state_dict(state)

    def save_for_predict(self, filename):
        """:492-509 - network weights without BERT / fixed embeddings, plus the config."""
        self.flush_readback()
        skip = ("CoVe", "ELMo", "AllenELMo", "Bert")
        state = {k: v for k, v in self.network.state_dict().items() if not k.startswith(skip)}
        for k in ("eval_embed.weight", "fixed_embedding_fast", "fixed_embedding_glove"):
            state.pop(k, None)
        cfg = {k: v for k, v in self.opt.items() if isinstance(v, (int, float, str, bool))}
        try:
            torch.save({"state_dict": {"network": state}, "config": cfg}, filename)
        except BaseException:
            log.info("[ WARN: Saving failed... continuing anyway. ]")
