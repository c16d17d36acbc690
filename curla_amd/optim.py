"""Adam over the learner's flat parameter buffers.

``FlatAdam`` IS a ``torch.optim.Adam`` (same constructor hyper-parameters,
``param_groups``, ``state`` keys, ``state_dict()`` / ``load_state_dict()``
format -- curl_sac.py:299-313 builds five of them), but its ``step()`` is one
HIP launch per contiguous run of live parameters in the flat buffer
(``curla_adam_step``) instead of torch's multi-tensor launches, which put ~70
workgroups on a 256-CU chip (45 us for 6 MB of parameters; the flat kernel
streams the same 7 arrays at HBM rate).  The moments live in two flat mirror
buffers; ``state[p]['exp_avg']`` / ``['exp_avg_sq']`` are views into them, so a
checkpoint written from ``state_dict()`` is what torch's Adam would write.
"""
import torch

from ._lib import call, stream


class FlatAdam(torch.optim.Adam):
    def __init__(self, params, flat, gflat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        """``params``: Parameters whose ``.data`` are views into ``flat`` and whose ``.grad`` (when not None) are the
        views at the same offsets of ``gflat``."""
        super().__init__(params, lr=lr, betas=betas, eps=eps)
        self._flat, self._gflat = flat, gflat
        self._plist = [p for g in self.param_groups for p in g["params"]]
        base = flat.data_ptr()
        self._span = []
        for p in self._plist:
            off = (p.data_ptr() - base) // 4
            if p.dtype != torch.float32 or off < 0 or off + p.numel() > flat.numel() or not p.is_contiguous():
                raise ValueError("FlatAdam: every parameter must be a contiguous float32 view into the flat buffer")
            self._span.append((off, off + p.numel()))
        self._lo = min(a for a, _ in self._span)
        self._hi = max(b for _, b in self._span)
        self._m = torch.zeros(self._hi - self._lo, device=flat.device, dtype=torch.float32)
        self._v = torch.zeros_like(self._m)
        self._steps = [0] * len(self._plist)
        self._plans = {}
        # Captured update graphs (CurlSacAgent.enable_update_graphs).  While a graph is being captured, ``_dyn`` is the
        # device address of this optimizer's two step-dependent floats (hyper_floats): the launch reads them from there
        # when it runs, and the step counts are NOT advanced by the launching code (the graph manager advances them
        # once per replay: advance()).
        self._dyn = None

    # -- state kept in torch's format ------------------------------------------------------------------------
    def _views(self, i):
        a, b = self._span[i]
        shape = self._plist[i].shape
        return self._m[a - self._lo:b - self._lo].view(shape), self._v[a - self._lo:b - self._lo].view(shape)

    def _ensure_state(self, i):
        p = self._plist[i]
        st = self.state[p]
        if len(st) == 0:
            m, v = self._views(i)
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"], st["exp_avg_sq"] = m, v
        return st

    def state_dict(self):
        for i, p in enumerate(self._plist):
            if p in self.state and len(self.state[p]):
                self.state[p]["step"] = torch.tensor(float(self._steps[i]), dtype=torch.float32)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)  # deep-copies the tensors: move them back into the flat mirrors
        with torch.no_grad():
            for i, p in enumerate(self._plist):
                st = self.state.get(p)
                if not st:
                    self._steps[i] = 0
                    continue
                m, v = self._views(i)
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
                st["exp_avg"], st["exp_avg_sq"] = m, v
                self._steps[i] = int(round(float(st["step"])))
                st["step"] = torch.tensor(float(self._steps[i]), dtype=torch.float32)

    # -- captured graphs: the step-dependent factors as data ------------------------------------------------------
    def live_indices(self, live=None):
        """Indices of the parameters a step touches: ``live`` (an iterable of Parameters) or all of them."""
        if live is None:
            return range(len(self._plist))
        ids = {id(p) for p in live}
        return [i for i, p in enumerate(self._plist) if id(p) in ids]

    def next_step(self, live=None):
        """The 1-based count of the step the next ``step()`` takes for the parameters it touches (they share it in
        graph mode; parameters whose gradient is None at step time -- the convs under detach_encoder -- are not
        ``live`` and keep their own count, as in torch's Adam)."""
        counts = {self._steps[i] for i in self.live_indices(live)}
        if len(counts) != 1:
            raise RuntimeError("FlatAdam: the parameters of a captured step must share one step count")
        return counts.pop() + 1

    def hyper_floats(self, t):
        """(lr / (1 - beta1^t), sqrt(1 - beta2^t)) as the two float32 values curla_adam_step evaluates from ``step`` on
        the host: the same double arithmetic (libm pow), rounded to float once."""
        import numpy as np
        g = self.param_groups[0]
        b1, b2 = float(g["betas"][0]), float(g["betas"][1])
        return np.float32(float(g["lr"]) / (1.0 - b1 ** float(t))), np.float32((1.0 - b2 ** float(t)) ** 0.5)

    def advance(self, live=None, by=1):
        """One step's worth of host bookkeeping (what step() does besides launching) for the parameters the step
        touches; ``by=-1`` takes it back (a capture that failed after the bookkeeping)."""
        for i in self.live_indices(live):
            self._ensure_state(i)
            self._steps[i] += by

    # -- the step --------------------------------------------------------------------------------------------
    def _plan(self, gi, group, live):
        """Contiguous runs [a, b) of the flat buffer covered by this group's live parameters (alignment padding
        between two adjacent parameters -- at most 3 floats, always zero gradient -- is bridged), each with the
        indices of its parameters."""
        idx0 = sum(len(g["params"]) for g in self.param_groups[:gi])
        items = sorted((self._span[idx0 + j] + (idx0 + j,) for j in range(len(group["params"])) if live[idx0 + j]))
        runs = []
        for a, b, i in items:
            if runs and a - runs[-1][1] <= 3 and a >= runs[-1][1]:
                runs[-1][1] = b
                runs[-1][2].append(i)
            else:
                runs.append([a, b, [i]])
        return runs

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        live = tuple(p.grad is not None for p in self._plist)
        plans = self._plans.get(live)
        if plans is None:
            base = self._gflat.data_ptr()
            for i, p in enumerate(self._plist):
                if live[i] and (p.grad.data_ptr() - base) // 4 != self._span[i][0]:
                    raise ValueError("FlatAdam: a gradient is not the flat gradient buffer's view of its parameter")
            plans = self._plans[live] = [self._plan(gi, g, live) for gi, g in enumerate(self.param_groups)]
        s = stream()
        for group, runs in zip(self.param_groups, plans):
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise NotImplementedError("FlatAdam implements the options the learner uses: plain Adam")
            lr, (b1, b2), eps = group["lr"], group["betas"], group["eps"]
            for a, b, members in runs:
                # parameters of one run normally share their step count; split the run where they do not
                k = 0
                while k < len(members):
                    t = self._steps[members[k]]
                    e = k
                    while e + 1 < len(members) and self._steps[members[e + 1]] == t:
                        e += 1
                    lo = a if k == 0 else self._span[members[k]][0]
                    hi = b if e == len(members) - 1 else self._span[members[e]][1]
                    if self._dyn is None:
                        for i in members[k:e + 1]:
                            self._ensure_state(i)
                            self._steps[i] = t + 1
                    elif len(plans) != 1 or len(runs) != 1 or e != len(members) - 1 or k != 0:
                        raise RuntimeError("FlatAdam: a captured step must be ONE run with one step count")
                    call("curla_adam_step", self._flat.data_ptr() + 4 * lo, self._gflat.data_ptr() + 4 * lo,
                         self._m.data_ptr() + 4 * (lo - self._lo), self._v.data_ptr() + 4 * (lo - self._lo), hi - lo,
                         float(lr), float(b1), float(b2), float(eps), t + 1, self._dyn, s)
                    k = e + 1
        return loss


    # -- two optimizers, one pass ------------------------------------------------------------------------------
    def _single_run(self):
        """(lo, hi, step, group) when every parameter is live, they form ONE run of the flat buffer and share their
        step count; else None."""
        if len(self.param_groups) != 1 or any(p.grad is None for p in self._plist):
            return None
        live = (True,) * len(self._plist)
        plans = self._plans.get(live)
        if plans is None:
            plans = self._plans[live] = [self._plan(0, self.param_groups[0], live)]
        runs = plans[0]
        if len(runs) != 1 or len(set(self._steps)) != 1:
            return None
        g = self.param_groups[0]
        if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
            return None
        return runs[0][0], runs[0][1], self._steps[0], g

    @staticmethod
    @torch.no_grad()
    def step_pair(first, second):
        """``first.step(); second.step()`` for two FlatAdams over the same flat buffer where first's parameters are the
        tail of second's (the encoder inside [CURL.W | encoder]): one launch that applies both steps per element in
        that order.  Anything else (partial gradients, unequal step counts, other layouts) takes the two steps."""
        plain = isinstance(first, FlatAdam) and isinstance(second, FlatAdam) and first._flat is second._flat and \
            first._gflat is second._gflat and "step" not in vars(first) and "step" not in vars(second)  # (a step
        # replaced on the instance -- tests do that to look at gradients before they are consumed -- is respected)
        ra, rb = (first._single_run(), second._single_run()) if plain else (None, None)
        if ra is None or rb is None or ra[1] != rb[1] or ra[0] < rb[0]:
            first.step()
            second.step()
            return
        (a0, a1, ta, ga), (b0, b1, tb, gb) = ra, rb
        if first._dyn is None:
            for opt in (first, second):
                for i in range(len(opt._plist)):
                    opt._ensure_state(i)
                    opt._steps[i] += 1
        elif second._dyn != first._dyn + 8:
            raise RuntimeError("FlatAdam.step_pair: the second optimizer's captured factors must follow the first's")
        flat, gflat = first._flat, first._gflat
        call("curla_adam_step2", flat.data_ptr() + 4 * b0, gflat.data_ptr() + 4 * b0,
             first._m.data_ptr() + 4 * (a0 - first._lo), first._v.data_ptr() + 4 * (a0 - first._lo),
             second._m.data_ptr() + 4 * (b0 - second._lo), second._v.data_ptr() + 4 * (b0 - second._lo), b1 - b0, a0 - b0,
             float(ga["lr"]), float(ga["betas"][0]), float(ga["betas"][1]), float(ga["eps"]), ta + 1,
             float(gb["lr"]), float(gb["betas"][0]), float(gb["betas"][1]), float(gb["eps"]), tb + 1, first._dyn, stream())


    # -- this optimizer's step and the target copy's soft update, one launch ---------------------------------------
    @staticmethod
    @torch.no_grad()
    def step_with_lerp(opt, target_flat, lo, hi, split, tau_a, tau_b):
        """``opt.step()`` followed by ``target[lo:hi] <- tau * flat[lo:hi] + (1 - tau) * target[lo:hi]`` (tau_a for
        [lo, lo + split), tau_b behind) in ONE launch, when ``opt`` is a FlatAdam whose live parameters are exactly the
        flat run [lo, hi) with one step count and a ``step`` that has not been replaced.  Returns False -- having done
        nothing -- when that does not hold: the caller then takes the two steps."""
        if not isinstance(opt, FlatAdam) or "step" in vars(opt):
            return False
        run = opt._single_run()
        # (the run ends with its last parameter, the announced block may include up to 3 floats of alignment padding
        # behind it: zeros in both copies, which a soft update leaves as they are)
        if run is None or run[0] != lo or not (run[1] <= hi <= run[1] + 3):
            return False
        _, hi, t, g = run
        if opt._dyn is None:
            for i in range(len(opt._plist)):
                opt._ensure_state(i)
                opt._steps[i] = t + 1
        call("curla_adam_step_lerp", opt._flat.data_ptr() + 4 * lo, opt._gflat.data_ptr() + 4 * lo,
             opt._m.data_ptr() + 4 * (lo - opt._lo), opt._v.data_ptr() + 4 * (lo - opt._lo), hi - lo, float(g["lr"]),
             float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), t + 1, opt._dyn, target_flat.data_ptr() + 4 * lo,
             split,
             float(tau_a), float(1.0 - tau_a), float(tau_b), float(1.0 - tau_b), stream())
        return True

    # -- this optimizer and a float64 scalar's, one launch -----------------------------------------------------------
    @staticmethod
    @torch.no_grad()
    def step_with_scalar(first, scalar_opt):
        """``first.step(); scalar_opt.step()`` where ``scalar_opt`` is a plain ``torch.optim.Adam`` over ONE float64
        one-element parameter on the device (log_alpha): its step rides in first's launch.  Anything else (first not a
        single flat run, a replaced ``step``, no gradient, other Adam options) takes the two steps."""
        groups = scalar_opt.param_groups
        p = groups[0]["params"][0] if len(groups) == 1 and len(groups[0]["params"]) == 1 else None
        g = groups[0] if p is not None else {}
        ok = isinstance(first, FlatAdam) and type(scalar_opt) is torch.optim.Adam and p is not None and \
            "step" not in vars(first) and "step" not in vars(scalar_opt) and p.dtype == torch.float64 and \
            p.numel() == 1 and p.is_cuda and p.grad is not None and p.grad.dtype == torch.float64 and \
            not (g.get("weight_decay", 0) or g.get("amsgrad") or g.get("maximize") or g.get("capturable") or
                 g.get("fused") or g.get("differentiable"))
        run = first._single_run() if ok else None
        if run is None:
            first.step()
            scalar_opt.step()
            return
        st = scalar_opt.state[p]
        if len(st) == 0:  # as torch's Adam initialises it (non-capturable: the step count lives on the host)
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
        elif st["step"].is_cuda:
            st["step"] = st["step"].cpu()
        t64 = int(round(float(st["step"]))) + 1
        lo, hi, t, grp = run
        frozen = first._dyn is not None  # (captured graph: the manager advances both step counts per replay)
        if not frozen:
            for i in range(len(first._plist)):
                first._ensure_state(i)
                first._steps[i] = t + 1
        call("curla_adam_step_scalar64", first._flat.data_ptr() + 4 * lo, first._gflat.data_ptr() + 4 * lo,
             first._m.data_ptr() + 4 * (lo - first._lo), first._v.data_ptr() + 4 * (lo - first._lo), hi - lo,
             float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]), t + 1, first._dyn,
             p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), float(g["lr"]),
             float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), t64, getattr(scalar_opt, "_curla_dyn64", None),
             stream())
        if not frozen:
            st["step"] = torch.tensor(float(t64), dtype=torch.float32)


__all__ = ["FlatAdam"]
