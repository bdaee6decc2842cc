"""L-BFGS with a strong-Wolfe line search for B INDEPENDENT problems in lock-step.

The reference's default branch (``core/config.py:29``, ``core/fitters/world_space.py:231-247``) hands every frame to
``torch.optim.LBFGS(params, max_iter=num_iters, lr=step_size, line_search_fn="strong_wolfe").step(closure)``.  On this
engine a closure call is an evaluate-only launch of the fit kernel, which costs the same for one frame as for thousands -
so for B independent frames (``use_previous_frame_init=False``, many sequences side by side) the per-frame optimisers are
run here as B state machines that advance together: ONE ``evaluate`` call per round serves the pending closure call of
every frame, and the optimiser's own arithmetic (two-loop recursion, cubic interpolation, bracket / zoom bookkeeping) is
vectorised over the frames with numpy instead of being B x hundreds of tiny tensor operations.

The algorithm is torch's (``torch/optim/lbfgs.py``: ``LBFGS.step``, ``_strong_wolfe``, ``_cubic_interpolate`` of the
installed torch 2.10), restated per frame: same constants (c1 1e-4, c2 0.9, tolerance_grad 1e-7, tolerance_change 1e-9,
history 100, ``max_eval = max_iter * 5 // 4``, ``max_ls = max_eval - evaluations so far``), same order of checks.  Vectors
are float32 like torch's flattened parameters, scalars float64.  It is NOT bit-identical to torch (dot products sum in a
different order), and L-BFGS at the reference's iteration counts is chaotic under rounding, so this
path is gated like the per-frame one: statistically against the reference's own envelope, plus agreement with
``torch.optim.LBFGS`` on well-conditioned problems (``tests/test_host_logic.py``).  ``WorldSpaceFitter`` keeps the
per-frame ``torch.optim.LBFGS`` for B = 1 (all the reference's API ever passes) and uses this for B > 1.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

_INIT, _BRACKET, _ZOOM, _DONE = 0, 1, 2, 3


def _rowdot(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Row-wise dot product in the vectors' precision (float32, as torch's ``Tensor.dot``), returned as float64."""
    return np.einsum("ij,ij->i", a, b).astype(np.float64)


def _cubic(x1, f1, g1, x2, f2, g2, lo=None, hi=None):
    """torch's ``_cubic_interpolate`` on arrays; bounds default to the interval [min(x1, x2), max(x1, x2)]."""
    if lo is None:
        lo, hi = np.minimum(x1, x2), np.maximum(x1, x2)
    with np.errstate(divide="ignore", invalid="ignore"):
        d1 = g1 + g2 - 3.0 * (f1 - f2) / (x1 - x2)
        sq = d1 * d1 - g1 * g2
        d2 = np.sqrt(np.where(sq >= 0, sq, 0.0))
        fwd = x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2.0 * d2))
        bwd = x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2.0 * d2))
    pos = np.where(x1 <= x2, fwd, bwd)
    # torch: min(max(pos, lo), hi) on Python floats, where max(a, b) is "b if b > a else a": a NaN position STAYS NaN (every
    # comparison with it is false), the step becomes NaN, the next loss is not finite and the frame stops - same here
    pos = np.where(lo > pos, lo, pos)
    pos = np.where(hi < pos, hi, pos)
    return np.where(sq >= 0, pos, 0.5 * (lo + hi))


class BatchedLBFGS:
    """``evaluate(X) -> (f, g)``: X (B, P) float32 -> losses (B,) and gradients (B, P); called once per round with the
    pending point of EVERY frame (frames that have finished are evaluated at their final point; the values are ignored)."""

    def __init__(self, evaluate: Callable[[np.ndarray], Tuple[np.ndarray, np.ndarray]], x0: np.ndarray, *, lr: float,
                 max_iter: int, max_eval: Optional[int] = None, tolerance_grad: float = 1e-7,
                 tolerance_change: float = 1e-9, history_size: int = 100):
        self.evaluate = evaluate
        self.vt = np.float64 if np.asarray(x0).dtype == np.float64 else np.float32     # vectors: float32 like torch's (float64: tests)
        self.x = np.array(x0, dtype=self.vt, copy=True)
        B, P = self.x.shape
        self.B, self.P = B, P
        self.lr, self.max_iter = float(lr), int(max_iter)
        self.max_eval = int(max_eval) if max_eval is not None else self.max_iter * 5 // 4
        self.tol_g, self.tol_c = float(tolerance_grad), float(tolerance_change)
        self.H = min(int(history_size), max(self.max_iter, 1))
        f32, f64 = self.vt, np.float64
        z = lambda *s, dt=f64: np.zeros(s, dtype=dt)
        self.phase = np.full(B, _INIT, dtype=np.int8)
        self.g, self.prev_g, self.d = z(B, P, dt=f32), z(B, P, dt=f32), z(B, P, dt=f32)
        self.loss, self.prev_loss = z(B), z(B)
        self.Y, self.S, self.RO = z(B, self.H, P, dt=f32), z(B, self.H, P, dt=f32), z(B, self.H)
        self.nold = np.zeros(B, dtype=np.int64)
        self.Hdiag = np.ones(B, dtype=f64)
        self.n_iter, self.evals = np.zeros(B, dtype=np.int64), np.zeros(B, dtype=np.int64)
        # line-search state
        self.t, self.t_prev, self.f_prev, self.gtd_prev = z(B), z(B), z(B), z(B)
        self.g_prev = z(B, P, dt=f32)
        self.f0, self.gtd0, self.d_norm = z(B), z(B), z(B)
        self.g0 = z(B, P, dt=f32)
        self.ls_iter, self.max_ls, self.ls_evals = (np.zeros(B, dtype=np.int64) for _ in range(3))
        self.first = np.zeros(B, dtype=bool)
        self.br_t, self.br_f, self.br_gtd = z(B, 2), z(B, 2), z(B, 2)
        self.br_g = z(B, 2, P, dt=f32)
        self.low = np.zeros(B, dtype=np.int64)
        self.insuf = np.zeros(B, dtype=bool)
        self.x_eval = self.x.copy()
        self.rounds = 0

    # ------------------------------------------------------------------------------------------------------------
    def run(self) -> np.ndarray:
        while np.any(self.phase != _DONE):
            f_new, g_new = self.evaluate(self.x_eval)
            self.rounds += 1
            self._advance(np.asarray(f_new, dtype=np.float64), np.asarray(g_new, dtype=self.vt))
        return self.x

    def _advance(self, f_new, g_new):
        ph = self.phase.copy()
        I = np.nonzero(ph == _INIT)[0]
        if I.size:
            self.loss[I], self.g[I], self.evals[I] = f_new[I], g_new[I], 1
            opt = np.abs(self.g[I]).max(axis=1) <= self.tol_g
            self.phase[I[opt]] = _DONE
            self._start_iteration(I[~opt])
        Bk = np.nonzero(ph == _BRACKET)[0]
        if Bk.size:
            self._bracket(Bk, f_new[Bk], g_new[Bk])
        Z = np.nonzero(ph == _ZOOM)[0]
        if Z.size:
            self._zoom_receive(Z, f_new[Z], g_new[Z])

    # ---- outer iteration (LBFGS.step) -----------------------------------------------------------------------------
    def _start_iteration(self, I):
        """Direction, step length and the first line-search evaluation of the next outer iteration for the frames `I`."""
        if not I.size:
            return
        self.n_iter[I] += 1
        first = self.n_iter[I] == 1
        F, N = I[first], I[~first]
        if F.size:
            self.d[F] = -self.g[F]
            self.nold[F] = 0
            self.Hdiag[F] = 1.0
            self.t[F] = np.minimum(1.0, 1.0 / np.abs(self.g[F]).sum(axis=1, dtype=self.vt).astype(np.float64)) * self.lr
        if N.size:
            y = self.g[N] - self.prev_g[N]
            s = self.d[N] * self.t[N, None].astype(self.vt)
            ys = _rowdot(y, s)
            upd = ys > 1e-10
            U = N[upd]
            if U.size:
                full = self.nold[U] == self.H
                if np.any(full):                                   # shift the history by one (limited memory)
                    W = U[full]
                    self.Y[W, :-1], self.S[W, :-1], self.RO[W, :-1] = self.Y[W, 1:], self.S[W, 1:], self.RO[W, 1:]
                    self.nold[W] -= 1
                k = self.nold[U]
                self.Y[U, k], self.S[U, k], self.RO[U, k] = y[upd], s[upd], 1.0 / ys[upd]
                self.nold[U] += 1
                self.Hdiag[U] = ys[upd] / _rowdot(y[upd], y[upd])
            # two-loop recursion; slots beyond a frame's own history hold ro = 0, so they drop out by themselves
            n = self.nold[N]
            top = int(n.max()) if n.size else 0
            q = -self.g[N]
            al = np.zeros((N.size, top), dtype=np.float64)
            for i in range(top - 1, -1, -1):
                live = i < n
                al[:, i] = np.where(live, _rowdot(self.S[N, i], q) * self.RO[N, i], 0.0)
                q = q - al[:, i, None].astype(self.vt) * self.Y[N, i]
            r = q * self.Hdiag[N, None].astype(self.vt)
            for i in range(top):
                live = i < n
                be = _rowdot(self.Y[N, i], r) * self.RO[N, i]
                r = r + np.where(live, al[:, i] - be, 0.0)[:, None].astype(self.vt) * self.S[N, i]
            self.d[N] = r
            self.t[N] = self.lr
        self.prev_g[I] = self.g[I]
        self.prev_loss[I] = self.loss[I]
        gtd = _rowdot(self.g[I], self.d[I])
        stop = ~(gtd <= -self.tol_c)                               # "gtd > -tolerance_change" (NaN stops too)
        self.phase[I[stop]] = _DONE
        self.x_eval[I[stop]] = self.x[I[stop]]
        L = I[~stop]
        if not L.size:
            return
        # strong-Wolfe line search from x along d: first evaluation at the initial step
        self.f0[L], self.gtd0[L], self.g0[L] = self.loss[L], gtd[~stop], self.g[L]
        self.d_norm[L] = np.abs(self.d[L]).max(axis=1)
        self.max_ls[L] = self.max_eval - self.evals[L]
        self.t_prev[L], self.f_prev[L], self.gtd_prev[L], self.g_prev[L] = 0.0, self.loss[L], gtd[~stop], self.g[L]
        self.ls_iter[L], self.ls_evals[L], self.first[L], self.insuf[L] = 0, 0, True, False
        self.phase[L] = _BRACKET
        self._issue(L)

    def _issue(self, L):
        self.x_eval[L] = self.x[L] + self.t[L, None].astype(self.vt) * self.d[L]
        self.ls_evals[L] += 1

    def _finish_line_search(self, L, t, f, g):
        """Line search over for the frames `L` (step t, loss f, gradient g there): take the step, run LBFGS.step's checks."""
        if not L.size:
            return
        self.t[L] = t
        self.x[L] = self.x[L] + t[:, None].astype(self.vt) * self.d[L]
        self.loss[L], self.g[L] = f, g
        self.evals[L] += self.ls_evals[L]
        opt = np.abs(g).max(axis=1) <= self.tol_g
        small_step = np.abs(self.d[L] * t[:, None].astype(self.vt)).max(axis=1) <= self.tol_c
        flat = np.abs(f - self.prev_loss[L]) < self.tol_c
        stop = (self.n_iter[L] >= self.max_iter) | (self.evals[L] >= self.max_eval) | opt | small_step | flat
        stop |= ~np.isfinite(f)
        self.phase[L[stop]] = _DONE
        self.x_eval[L[stop]] = self.x[L[stop]]
        self._start_iteration(L[~stop])

    # ---- _strong_wolfe: bracket phase -----------------------------------------------------------------------------
    def _bracket(self, Bk, f_new, g_new):
        c1, c2 = 1e-4, 0.9
        self.ls_iter[Bk] += ~self.first[Bk]                        # the first evaluation precedes the loop
        self.first[Bk] = False
        t, f0, gtd0 = self.t[Bk], self.f0[Bk], self.gtd0[Bk]
        gtd_new = _rowdot(g_new, self.d[Bk])
        exhausted = self.ls_iter[Bk] >= self.max_ls[Bk]            # "ls_iter == max_ls": bracket = [0, t], no zoom
        armijo = (f_new > f0 + c1 * t * gtd0) | ((self.ls_iter[Bk] > 1) & (f_new >= self.f_prev[Bk]))
        wolfe = ~armijo & (np.abs(gtd_new) <= -c2 * gtd0)
        uphill = ~armijo & ~wolfe & (gtd_new >= 0)
        to_zoom = ~exhausted & (armijo | uphill)
        done = ~exhausted & wolfe
        cont = ~exhausted & ~armijo & ~wolfe & ~uphill
        if np.any(exhausted):
            E = Bk[exhausted]
            lower0 = f0[exhausted] <= f_new[exhausted]
            self._finish_line_search(E, np.where(lower0, 0.0, t[exhausted]), np.where(lower0, f0[exhausted], f_new[exhausted]),
                                     np.where(lower0[:, None], self.g0[E], g_new[exhausted]))
        if np.any(done):
            self._finish_line_search(Bk[done], t[done], f_new[done], g_new[done])
        if np.any(to_zoom):
            Zn = Bk[to_zoom]
            self.br_t[Zn] = np.stack([self.t_prev[Zn], t[to_zoom]], axis=1)
            self.br_f[Zn] = np.stack([self.f_prev[Zn], f_new[to_zoom]], axis=1)
            self.br_gtd[Zn] = np.stack([self.gtd_prev[Zn], gtd_new[to_zoom]], axis=1)
            self.br_g[Zn, 0], self.br_g[Zn, 1] = self.g_prev[Zn], g_new[to_zoom]
            self.low[Zn] = np.where(self.br_f[Zn, 0] <= self.br_f[Zn, 1], 0, 1)
            self.insuf[Zn] = False
            self.phase[Zn] = _ZOOM
            self._zoom_next(Zn)
        if np.any(cont):
            C = Bk[cont]
            tc, tp = t[cont], self.t_prev[C]
            t_next = _cubic(tp, self.f_prev[C], self.gtd_prev[C], tc, f_new[cont], gtd_new[cont],
                            lo=tc + 0.01 * (tc - tp), hi=tc * 10.0)
            self.t_prev[C], self.f_prev[C], self.gtd_prev[C], self.g_prev[C] = tc, f_new[cont], gtd_new[cont], g_new[cont]
            self.t[C] = t_next
            self._issue(C)

    # ---- _strong_wolfe: zoom phase --------------------------------------------------------------------------------
    def _zoom_next(self, Z):
        """Loop head of the zoom phase for the frames `Z`: stop (budget, bracket too small) or pick the next trial step."""
        if not Z.size:
            return
        bt = self.br_t[Z]
        width = np.abs(bt[:, 1] - bt[:, 0])
        stop = (self.ls_iter[Z] >= self.max_ls[Z]) | (width * self.d_norm[Z] < self.tol_c)
        if np.any(stop):
            S = Z[stop]
            lo = self.low[S]
            ar = np.arange(S.size)
            self._finish_line_search(S, self.br_t[S][ar, lo], self.br_f[S][ar, lo], self.br_g[S][ar, lo])
        G = Z[~stop]
        if not G.size:
            return
        bt, bf, bg = self.br_t[G], self.br_f[G], self.br_gtd[G]
        t = _cubic(bt[:, 0], bf[:, 0], bg[:, 0], bt[:, 1], bf[:, 1], bg[:, 1])
        hi, lo = bt.max(axis=1), bt.min(axis=1)
        eps = 0.1 * (hi - lo)
        near = np.minimum(hi - t, t - lo) < eps
        move = near & (self.insuf[G] | (t >= hi) | (t <= lo))
        t = np.where(move, np.where(np.abs(t - hi) < np.abs(t - lo), hi - eps, lo + eps), t)
        self.insuf[G] = near & ~move
        self.t[G] = t
        self._issue(G)

    def _zoom_receive(self, Z, f_new, g_new):
        c1, c2 = 1e-4, 0.9
        self.ls_iter[Z] += 1
        t, f0, gtd0 = self.t[Z], self.f0[Z], self.gtd0[Z]
        gtd_new = _rowdot(g_new, self.d[Z])
        ar = np.arange(Z.size)
        low = self.low[Z]
        high = 1 - low
        bt, bf, bgtd, bg = self.br_t[Z], self.br_f[Z], self.br_gtd[Z], self.br_g[Z]
        worse = (f_new > f0 + c1 * t * gtd0) | (f_new >= bf[ar, low])
        # Armijo violated or not below the lowest point: the trial replaces the HIGH end
        w = np.nonzero(worse)[0]
        bt[w, high[w]], bf[w, high[w]], bgtd[w, high[w]], bg[w, high[w]] = t[w], f_new[w], gtd_new[w], g_new[w]
        # otherwise the trial becomes the LOW end (and the old low the high end when the slope says so)
        b = np.nonzero(~worse)[0]
        wolfe = np.zeros(Z.size, dtype=bool)
        if b.size:
            wolfe[b] = np.abs(gtd_new[b]) <= -c2 * gtd0[b]
            flip = b[~wolfe[b] & (gtd_new[b] * (bt[b, high[b]] - bt[b, low[b]]) >= 0)]
            bt[flip, high[flip]], bf[flip, high[flip]] = bt[flip, low[flip]], bf[flip, low[flip]]
            bgtd[flip, high[flip]], bg[flip, high[flip]] = bgtd[flip, low[flip]], bg[flip, low[flip]]
            bt[b, low[b]], bf[b, low[b]], bgtd[b, low[b]], bg[b, low[b]] = t[b], f_new[b], gtd_new[b], g_new[b]
        self.br_t[Z], self.br_f[Z], self.br_gtd[Z], self.br_g[Z] = bt, bf, bgtd, bg
        new_low = np.where(bf[:, 0] <= bf[:, 1], 0, 1)
        self.low[Z] = np.where(worse, new_low, low)
        if np.any(wolfe):
            D = Z[wolfe]
            self._finish_line_search(D, t[wolfe], f_new[wolfe], g_new[wolfe])
        self._zoom_next(Z[~wolfe])


def minimize(evaluate, x0, *, lr: float, max_iter: int, **kw) -> Tuple[np.ndarray, int]:
    """Run the B optimisers to the end; returns (final points (B, P) float32, rounds = evaluate calls made)."""
    opt = BatchedLBFGS(evaluate, x0, lr=lr, max_iter=max_iter, **kw)
    x = opt.run()
    return x, opt.rounds
