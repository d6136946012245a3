"""Asynchronous scalar log channel of the train loop.

The reference pulls five scalars to the host EVERY step with `.item()` (ivae_ardae.py:756-758,774,837-841) and, every
`--log-interval` steps, prints one line (`:858-891`, utils/msc.py:117-125 -> `log.txt`) and writes twelve tensorboard scalars
(`:894-906`).  Here the fused step appends one record to a device ring buffer (`ardae_log_scalars`, the last kernel of the
step: captured in the step's HIP graph, no host synchronisation); `drain()` reads the records written since the previous
drain over a side stream into pinned host memory, without stalling the compute stream, and emits

  * `log.txt`        -- the reference's line format, one line per step record
  * `scalars.jsonl`  -- {"iter": i, "<train_mode>/model/loss/step": ..., ...} with the reference's twelve tensorboard tags
                        (any tensorboard writer can replay them: `writer.add_scalar(tag, value, iter)`)
"""
import json
import os
import time

import numpy as np
import torch

from . import _lib as L

TAGS = ("model/loss", "model/recon", "model/prior", "model/beta", "cdae/loss", "cdae/std/eff/mean", "cdae/std/true/mean",
        "cdae/std/eff/max", "cdae/std/true/max", "cdae/std/eff/min", "cdae/std/true/min", "cdae/lr")     # ivae_ardae.py:894-906


class ScalarLog:
    def __init__(self, engine, capacity=1024, path=None, train_mode="train", iters_per_epoch=None):
        self.eng, self.capacity, self.path, self.train_mode = engine, int(capacity), path, train_mode
        self.iters_per_epoch = iters_per_epoch
        self.ring = torch.zeros(self.capacity, L.LOG_RECORD_FLOATS, device=engine.dev, dtype=torch.float32)
        self._host = torch.zeros(self.capacity, L.LOG_RECORD_FLOATS, dtype=torch.float32).pin_memory()
        self._stream = torch.cuda.Stream(device=engine.dev)
        self._next = engine.step_count + 1          # first iteration this log has not reported yet
        self.dropped = 0                            # records overwritten before they were drained
        self._t_last = time.time()
        engine.attach_log(self)

    def resync(self):
        """The engine's step count was set from outside (load_checkpoints): records older than it are stale, the next one is new."""
        self._next = self.eng.step_count + 1

    def record(self, beta):
        """Called by the engine as the last launch of a step (on the step's stream)."""
        e = self.eng
        L.check(e.lib.ardae_log_scalars(L.ptr(e.loss_c), L.ptr(e.losses_m), L.ptr(e.std_b), e.B, float(beta), float(e.cfg.d_lr),
                                        e.state.data_ptr(), L.ptr(self.ring), self.capacity, L.stream_ptr()), "ardae_log_scalars")

    def drain(self):
        """Records of the steps finished since the last drain (oldest first); appends them to log.txt / scalars.jsonl."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self._stream):
            self._stream.wait_event(ev)
            self._host.copy_(self.ring, non_blocking=True)
        self._stream.synchronize()                  # waits for the copy only; the compute stream keeps running
        raw = self._host.numpy()
        ints = raw.view(np.int32)
        iters = ints[:, 0].astype(np.int64) + (ints[:, 10].astype(np.int64) << 31)
        order = [i for i in np.argsort(iters) if iters[i] >= self._next]
        if not order:
            return []
        newest = int(iters[order[-1]])
        self.dropped += (newest - self._next + 1) - len(order)
        now = time.time()
        ms = 1e3 * (now - self._t_last) / max(1, newest - self._next + 1)
        self._t_last = now
        s = float(self.eng.cfg.std_scale)
        out = []
        for i in order:
            r = raw[i]
            vals = {"model/loss": r[1], "model/recon": r[2], "model/prior": r[3], "model/beta": r[4], "cdae/loss": r[5],
                    "cdae/std/eff/mean": r[6], "cdae/std/true/mean": r[6] / s, "cdae/std/eff/max": r[7], "cdae/std/true/max": r[7] / s,
                    "cdae/std/eff/min": r[8], "cdae/std/true/min": r[8] / s, "cdae/lr": r[9]}
            rec = {"iter": int(iters[i]), "ms_per_step": ms}
            rec.update({f"{self.train_mode}/{t}/step": float(vals[t]) for t in TAGS})
            out.append(rec)
        self._next = newest + 1
        if self.path is not None:
            os.makedirs(self.path, exist_ok=True)
            with open(os.path.join(self.path, "log.txt"), "a+") as f:           # utils/msc.py:117-125
                for rec in out:
                    f.write(self.format_line(rec) + "\n")
            with open(os.path.join(self.path, "scalars.jsonl"), "a+") as f:
                for rec in out:
                    f.write(json.dumps(rec) + "\n")
        return out

    def format_line(self, rec):
        """The reference's log line (ivae_ardae.py:858-891)."""
        g = lambda t: rec[f"{self.train_mode}/{t}/step"]
        it = rec["iter"]
        per = self.iters_per_epoch
        epoch, idx, tot = ((it - 1) // per + 1, (it - 1) % per + 1, per) if per else (0, it, 0)
        return ('| iter {:d} | epoch {:3d} | {:5d}/{:5d} | ms/step {:5.2f} | dlr {:.5f} | (eff) std {:5.3f} | (true) std {:5.3f} '
                '| (eff) max std {:5.3f} | (eff) min std {:5.3f} | beta {:5.3f} | loss (vae) {:5.3f} | loss (recon) {:5.3f} '
                '| loss (prior) {:5.3f} | loss (cdae) {:5.4f} ').format(
                    it, epoch, idx, tot, rec["ms_per_step"], g("cdae/lr"), g("cdae/std/eff/mean"), g("cdae/std/true/mean"),
                    g("cdae/std/eff/max"), g("cdae/std/eff/min"), g("model/beta"), g("model/loss"), g("model/recon"), g("model/prior"),
                    g("cdae/loss"))
