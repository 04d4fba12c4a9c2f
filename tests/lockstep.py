"""Lock-step parity harness: runs the product engine on CPU with the torch
per-kernel specs and, for every single op, the HIP kernel on a GPU mirror of
the exact same inputs, comparing every buffer afterwards.  Errors therefore
localise to one kernel and never propagate."""
import torch

from oracle.kernel_spec import SpecBackend

FINE_OPS = ["logmel", "conv1", "gemm", "copy_rows", "layernorm", "log_softmax_rows", "block_pack",
            "ctx_handoff", "enc_attention", "ctc_extend_state", "dec_embed", "dec_self_attn",
            "dec_cross_attn", "logsoftmax_topk", "ctc_prefix_scan", "fuse_topw", "beam_prune",
            "ctc_gather_state"]


class LockstepBackend(SpecBackend):
    name = "lockstep"

    def __init__(self, hip, atol=2e-4, rtol=2e-4):
        self.hip = hip
        self.atol, self.rtol = atol, rtol
        self.sb_cpu = None
        self.sb_gpu = None
        self.report = {}       # op -> max abs diff seen
        self.failures = []
        self.int_mismatch = []

    def attach(self, sb_cpu, sb_gpu):
        self.sb_cpu, self.sb_gpu = sb_cpu, sb_gpu
        self._names = [k for k, v in vars(sb_cpu).items() if isinstance(v, torch.Tensor)]
        self._cpu_ptr = {getattr(sb_cpu, k).data_ptr(): k for k in self._names}
        self._wmap = {}
        wc, wg = sb_cpu.w, sb_gpu.w
        for k, v in vars(wc).items():
            if isinstance(v, torch.Tensor):
                self._wmap[v.data_ptr()] = getattr(wg, k)
        for lc, lg in list(zip(wc.enc, wg.enc)) + list(zip(wc.dec, wg.dec)):
            for k, v in lc.items():
                self._wmap[v.data_ptr()] = lg[k]

    def _sync_to_gpu(self):
        for k in self._names:
            getattr(self.sb_gpu, k).copy_(getattr(self.sb_cpu, k))

    def _xlate(self, a):
        if a is self.sb_cpu:
            return self.sb_gpu
        if a is self.sb_cpu.w:
            return self.sb_gpu.w
        if isinstance(a, torch.Tensor):
            p = a.data_ptr()
            if p in self._cpu_ptr and a.numel() == getattr(self.sb_cpu, self._cpu_ptr[p]).numel():
                return getattr(self.sb_gpu, self._cpu_ptr[p])
            if p in self._wmap:
                return self._wmap[p]
            return a.to(self.sb_gpu.dev)
        return a

    def _compare(self, op):
        worst = 0.0
        for k in self._names:
            c = getattr(self.sb_cpu, k)
            g = getattr(self.sb_gpu, k).cpu()
            if c.dtype in (torch.int32, torch.int64):
                if not torch.equal(c, g):
                    self.int_mismatch.append((op, k, int((c != g).sum())))
                continue
            c64, g64 = c.double(), g.double()
            if torch.isnan(g64).any() and not torch.isnan(c64).any():
                self.failures.append((op, k, "nan"))
                continue
            diff = (c64 - g64).abs()
            diff[c64 == g64] = 0.0     # identical sentinels / infinities
            tol = self.atol + self.rtol * c64.abs()
            bad = diff > tol
            if bad.any():
                self.failures.append((op, k, float(diff[bad].max()), int(bad.sum())))
            else:
                worst = max(worst, float(diff.max()) if diff.numel() else 0.0)
        self.report[op] = max(self.report.get(op, 0.0), worst)

    def _both(self, op, args, kwargs):
        self._sync_to_gpu()
        gargs = [self._xlate(a) for a in args]
        gkw = {k: self._xlate(v) for k, v in kwargs.items()}
        getattr(self.hip, op)(*gargs, **gkw)
        torch.cuda.synchronize()
        getattr(SpecBackend, op)(self, *args, **kwargs)
        self._compare(op)


def _make(op):
    def f(self, *args, **kwargs):
        return self._both(op, args, kwargs)
    f.__name__ = op
    return f


for _op in FINE_OPS:
    setattr(LockstepBackend, _op, _make(_op))
