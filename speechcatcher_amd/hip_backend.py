"""Backend that runs every engine op as hand-written HIP kernels through the
C ABI of libscasr.so (include/scasr.h).  torch is used only as the owner of
device memory and of the HIP stream; every pointer handed to the library is a
raw device address."""
import ctypes as C
import os

import torch

from . import _abi


def _p(t):
    return 0 if t is None else t.data_ptr()


class HipBackend:
    name = "hip"

    def __init__(self, device="cuda:0", use_graphs=True):
        if not torch.cuda.is_available():
            raise _abi.ScasrError("HipBackend needs a ROCm GPU (torch.cuda.is_available() is False)")
        self.lib = _abi.load()
        self.device = torch.device(device)
        self.use_graphs = use_graphs
        self._enc_graphs = {}
        # split-K partial sums of sc_gemm live in this caller-owned workspace
        self.workspace = torch.empty(128 << 20, dtype=torch.uint8, device=self.device)
        self._chk(self.lib.sc_set_workspace(self.workspace.data_ptr(), self.workspace.numel()), "sc_set_workspace")

    def bind_stream(self, stream):
        """Give the batch that runs on `stream` its own split-K workspace, so that
        several batches can run concurrently on different HIP streams."""
        ws = torch.empty(128 << 20, dtype=torch.uint8, device=self.device)
        self._stream_ws = getattr(self, "_stream_ws", [])
        self._stream_ws.append(ws)
        self._chk(self.lib.sc_set_stream_workspace(stream.cuda_stream, ws.data_ptr(), ws.numel()),
                  "sc_set_stream_workspace")

    def check_supported(self, cfg, beam_size):
        """Reject, at construction time, model dimensions the decode kernels have no instantiation for
        (the first decode step would otherwise fail in the middle of a stream)."""
        dk = cfg.d_model // cfg.dec_heads
        if cfg.d_model % cfg.dec_heads or dk not in (16, 32, 64) or (dk == 64 and beam_size > 10):
            raise _abi.ScasrError(
                f"decoder head dim {dk} (d_model {cfg.d_model} / {cfg.dec_heads} heads) with beam {beam_size} is not "
                "supported: the HIP decoder attention kernels are instantiated for head dims 16, 32 and 64 (64: beam <= 10; "
                "csrc/search.hip)")
        ek = cfg.d_model // cfg.enc_heads
        if cfg.d_model % cfg.enc_heads or ek not in (16, 32, 64):
            raise _abi.ScasrError(f"encoder head dim {ek} is not supported (16, 32 or 64)")
        if beam_size > 16:
            raise _abi.ScasrError(f"beam size {beam_size} > 16 is not supported by the decoder attention kernels")
        if cfg.d_model % 32:
            raise _abi.ScasrError("d_model must be a multiple of 32 (sc_gemm K tiles)")

    # ------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _chk(self, rc, what):
        if rc != 0:
            _abi.check(rc, what)

    # ------------------------------------------------------------------
    def logmel(self, w, pcm, pcap, jobs, n_jobs, max_keep, featbuf):
        cfg = w.cfg
        mode = 0 if not w.has_mvn else (2 if w.mvn_is_f64 else 1)
        self._chk(self.lib.sc_logmel(_p(pcm), pcap, _p(jobs), n_jobs, max_keep, _p(w.window), _p(w.mel_fb),
                                     _p(w.twiddle), _p(w.mean64), _p(w.std64), mode, cfg.n_fft,
                                     cfg.hop_length, cfg.win_length, cfg.n_mels, _p(featbuf),
                                     self._stream()), "sc_logmel")

    def conv1(self, w, featbuf, jobs, n_jobs, max_t1, c1):
        cfg = w.cfg
        self._chk(self.lib.sc_conv1(_p(featbuf), cfg.n_mels, _p(jobs), n_jobs, max_t1, _p(w.conv1_w),
                                    _p(w.conv1_b), cfg.d_model, _p(c1), self._stream()), "sc_conv1")

    def gemm(self, A, a_rows, lda, W, bias, Cm, c_rows, ldc, M, N, K, relu=False, conv_f1=0,
             residual=False, naive=False, split16=False):
        flags = (1 if relu else 0) | (2 if residual else 0) | (4 if naive else 0) | (16 if split16 else 0)
        self._chk(self.lib.sc_gemm(_p(A), _p(a_rows), lda, _p(W), _p(bias), _p(Cm), _p(c_rows), ldc,
                                   M, N, K, flags, conv_f1, self._stream()), "sc_gemm")

    def gemm_ln(self, A, a_rows, lda, W, bias, Cm, c_rows, ldc, M, N, K, ln_g, ln_b, ln_out,
                relu=False, conv_f1=0, residual=False, eps=1e-12, ln_at_crows=False):
        flags = (1 if relu else 0) | (2 if residual else 0) | (8 if ln_at_crows else 0)
        self._chk(self.lib.sc_gemm_ln(_p(A), _p(a_rows), lda, _p(W), _p(bias), _p(Cm), _p(c_rows), ldc,
                                      M, N, K, flags, conv_f1, _p(ln_g), _p(ln_b), eps, _p(ln_out),
                                      ln_out.shape[-1], self._stream()), "sc_gemm_ln")

    def proj_ln_proj(self, A, lda, W1, b1, X, ldx, ln_g, ln_b, XN, W2, b2, Q, M, D, eps=1e-12, rows=None):
        self._chk(self.lib.sc_proj_ln_proj(_p(A), lda, _p(W1), _p(b1), _p(X), ldx, _p(ln_g), _p(ln_b), eps,
                                           _p(XN), D, _p(W2), _p(b2), _p(Q), D, _p(rows), M, D, self._stream()),
                  "sc_proj_ln_proj")

    def ffn_ln(self, XN, rows, M, D, F, W1p, b1, W2p, b2, X, ln_g, ln_b, ln_out, eps=1e-12):
        self._chk(self.lib.sc_ffn_ln(_p(XN), _p(rows), M, D, F, _p(W1p), _p(b1), _p(W2p), _p(b2), _p(X),
                                     _p(ln_g), _p(ln_b), eps, _p(ln_out), self._stream()), "sc_ffn_ln")

    def rowtile_proj_h(self, A, M, D, Wh, bias, N, C, ln_g=None, ln_b=None, R=None, g2=None, b2=None, LN2=None, eps=1e-12):
        """sc_rowtile_proj with fp16 weights (the packed fragments as torch.float16): fp16 MFMA inputs, fp32 accumulation"""
        assert Wh.dtype == torch.float16
        self._chk(self.lib.sc_rowtile_proj_h(_p(A), A.shape[-1], M, D, _p(ln_g), _p(ln_b), eps, _p(Wh), _p(bias), N, _p(R), _p(C),
                                             C.shape[-1], _p(g2), _p(b2), _p(LN2), self._stream()), "sc_rowtile_proj_h")

    def rowtile_proj_s(self, A, M, D, Ws, bias, N, C, ln_g=None, ln_b=None, R=None, g2=None, b2=None, LN2=None, eps=1e-12):
        """sc_rowtile_proj with the fp16 hi | lo split of the weights (weights.split_panel_weight): fp32-grade results"""
        assert Ws.dtype == torch.float16
        self._chk(self.lib.sc_rowtile_proj_s(_p(A), A.shape[-1], M, D, _p(ln_g), _p(ln_b), eps, _p(Ws), _p(bias), N, _p(R), _p(C),
                                             C.shape[-1], _p(g2), _p(b2), _p(LN2), self._stream()), "sc_rowtile_proj_s")

    def ffn_ln_h(self, XN, rows, M, D, F, W1h, b1, W2h, b2, X, ln_g, ln_b, ln_out, eps=1e-12):
        """sc_ffn_ln with fp16 weights (the packed fragments as torch.float16): fp16 MFMA inputs, fp32 accumulation"""
        assert W1h.dtype == torch.float16 and W2h.dtype == torch.float16
        self._chk(self.lib.sc_ffn_ln_h(_p(XN), _p(rows), M, D, F, _p(W1h), _p(b1), _p(W2h), _p(b2), _p(X),
                                       _p(ln_g), _p(ln_b), eps, _p(ln_out), self._stream()), "sc_ffn_ln_h")

    def ffn_ln_s(self, XN, rows, M, D, F, W1s, b1, W2s, b2, X, ln_g, ln_b, ln_out, eps=1e-12):
        """sc_ffn_ln with the fp16 hi | lo split of the weights (weights.split_panel_weight): fp32-grade results from
        fp16 MFMA inputs"""
        assert W1s.dtype == torch.float16 and W2s.dtype == torch.float16
        self._chk(self.lib.sc_ffn_ln_s(_p(XN), _p(rows), M, D, F, _p(W1s), _p(b1), _p(W2s), _p(b2), _p(X),
                                       _p(ln_g), _p(ln_b), eps, _p(ln_out), self._stream()), "sc_ffn_ln_s")

    def ffn_ln_proj(self, XN, rows, M, D, F, W1p, b1, W2p, b2, Xin, Xout, ln_g, ln_b, Wq, bq, Q, N, eps=1e-12):
        self._chk(self.lib.sc_ffn_ln_proj(_p(XN), _p(rows), M, D, F, _p(W1p), _p(b1), _p(W2p), _p(b2), _p(Xin),
                                          _p(Xout), _p(ln_g), _p(ln_b), eps, None, _p(Wq), _p(bq), _p(Q), N,
                                          self._stream()), "sc_ffn_ln_proj")

    def rowtile_proj(self, A, M, D, Wp, bias, N, C_out, ln_g=None, ln_b=None, R=None, g2=None, b2=None, LN2=None,
                     eps=1e-12):
        """C = [LN](A) . W^T + bias [+ R] [-> LN2]  (include/scasr.h: sc_rowtile_proj)"""
        self._chk(self.lib.sc_rowtile_proj(_p(A), A.shape[-1], M, D, _p(ln_g), _p(ln_b), eps, _p(Wp), _p(bias), N,
                                           _p(R), _p(C_out), C_out.shape[-1], _p(g2), _p(b2), _p(LN2),
                                           self._stream()), "sc_rowtile_proj")

    def kv_rows_to_half(self, stage, rows, m, n_layers, tcap, d2, ckv):
        self._chk(self.lib.sc_kv_rows_to_half(_p(stage), _p(rows), m, n_layers, tcap, d2, _p(ckv), self._stream()),
                  "sc_kv_rows_to_half")

    def copy_rows(self, src, src_rows, dst, dst_rows, n, width):
        self._chk(self.lib.sc_copy_rows(_p(src), _p(src_rows), _p(dst), _p(dst_rows), n, width,
                                        self._stream()), "sc_copy_rows")

    def layernorm(self, src, src_rows, dst, dst_rows, M, g, b, eps=1e-12):
        self._chk(self.lib.sc_layernorm(_p(src), _p(src_rows), src.shape[-1], _p(dst), _p(dst_rows),
                                        dst.shape[-1], M, g.numel(), _p(g), _p(b), eps,
                                        self._stream()), "sc_layernorm")

    def log_softmax_rows(self, x, rows, n, V):
        self._chk(self.lib.sc_log_softmax_rows(_p(x), _p(rows), n, V, self._stream()), "sc_log_softmax_rows")

    def block_pack(self, w, subbuf, jobs, nb, R, xblk):
        self._chk(self.lib.sc_block_pack(_p(subbuf), _p(jobs), nb, R, _p(w.pe), w.cfg.d_model, _p(xblk),
                                         self._stream()), "sc_block_pack")

    def ctx_handoff(self, x, R, jobs, ns, state, layer):
        self._chk(self.lib.sc_ctx_handoff(_p(x), R, _p(jobs), ns, _p(state), layer, x.shape[-1],
                                          self._stream()), "sc_ctx_handoff")

    def enc_attention(self, qkv, att, nblk, R, H, masked):
        self._chk(self.lib.sc_enc_attention(_p(qkv), _p(att), nblk, R, H, att.shape[-1], int(masked),
                                            self._stream()), "sc_enc_attention")

    def glu_dwconv_bn_swish(self, y, B, T, Cc, ksize, dw_w, dw_b, bn_g, bn_b, bn_mean, bn_var, eps, out):
        self._chk(self.lib.sc_glu_dwconv_bn_swish(_p(y), B, T, Cc, ksize, _p(dw_w), _p(dw_b), _p(bn_g), _p(bn_b),
                                                  _p(bn_mean), _p(bn_var), eps, _p(out), self._stream()),
                  "sc_glu_dwconv_bn_swish")

    def relpos_attention(self, qkv, p, bias_u, bias_v, out, B, T, H, mask=None):
        """mask: None, uint8 [B][T] (keys) or [B][T][T]; 0 = masked out"""
        mode = 0 if mask is None else (1 if mask.dim() == 2 else 2)
        self._chk(self.lib.sc_relpos_attention_masked(_p(qkv), _p(p), _p(bias_u), _p(bias_v), _p(out), B, T, H,
                                                      out.shape[-1], _p(mask), mode, self._stream()),
                  "sc_relpos_attention_masked")

    def _enc_layer_table(self, w):
        # cached ON the weights object (never keyed by id(): ids are recycled)
        arr = getattr(w, "_sc_enc_layer_table", None)
        if arr is None:
            arr = (_abi.EncLayer * len(w.enc))()
            for i, lw in enumerate(w.enc):
                for name, _ in _abi.EncLayer._fields_:
                    setattr(arr[i], name, lw[name].data_ptr() if name in lw else None)   # w1_h / w2_h: optional
            w._sc_enc_layer_table = arr
        return arr

    def encoder_layers(self, w, x, nblk, R, masked, jobs, ns, past_ctx, xn, qkv, att, ffh):
        """All encoder layers.  The ~360 launches are replayed from a hipGraph
        keyed by every argument that shapes the launch sequence (pointers
        included, so a cached graph is only ever replayed with the exact
        arguments it was captured with)."""
        cfg = w.cfg
        tab = self._enc_layer_table(w)
        st = self._stream()

        def launch():
            self._chk(self.lib.sc_encoder_layers(C.cast(tab, C.c_void_p), len(w.enc), _p(x), nblk, R, int(masked),
                                                 _p(jobs), ns, _p(past_ctx), _p(xn), _p(qkv), _p(att), _p(ffh),
                                                 cfg.d_model, cfg.enc_heads, cfg.ffn_dim, cfg.ln_eps,
                                                 st), "sc_encoder_layers")

        if not self.use_graphs or st == 0:
            return launch()
        key = (C.addressof(tab), _p(x), nblk, R, int(masked), _p(jobs), ns, _p(past_ctx), _p(xn), _p(qkv),
               _p(att), _p(ffh), st)
        g = self._enc_graphs.get(key)
        if g is None:
            if len(self._enc_graphs) >= 16:      # ragged callers: do not hoard graphs
                return launch()
            self._chk(self.lib.sc_graph_capture_begin(st), "sc_graph_capture_begin")
            try:
                launch()
            finally:
                out = C.c_void_p()
                rc = self.lib.sc_graph_capture_end(st, C.byref(out))
            self._chk(rc, "sc_graph_capture_end")
            self._enc_graphs[key] = g = out
        self._chk(self.lib.sc_graph_launch(g, st), "sc_graph_launch")

    # ------------------------------------------------------------------
    def search_struct(self, sb):
        cached = getattr(sb, "_sc_search_struct", None)
        if cached is not None:
            return cached[0]
        w, cfg = sb.w, sb.cfg
        layers = (_abi.DecLayer * len(w.dec))()
        for i, lw in enumerate(w.dec):
            for name, _ in _abi.DecLayer._fields_:
                setattr(layers[i], name, lw[name].data_ptr() if name in lw else None)
        s = _abi.Search()
        s.S, s.W, s.K, s.V, s.d, s.H, s.F = sb.S, sb.W, sb.K, cfg.vocab_size, cfg.d_model, cfg.dec_heads, cfg.ffn_dim
        s.n_layers, s.TCAP, s.LCAP = cfg.dec_layers, sb.TCAP, sb.LCAP
        s.blank, s.eos, s.sos = cfg.blank_id, cfg.eos_id, cfg.sos_id
        s.w_dec, s.w_ctc, s.ln_eps = sb.search.decoder_weight, sb.search.ctc_weight, cfg.ln_eps
        for name in ("ctrl", "flags", "ctcx", "ckv", "skv", "yseq", "xpos", "anc", "score", "sc_dec",
                     "sc_ctc", "ctc_r", "ctc_rs", "ctc_s", "ctc_rnew", "dx", "dxn", "dqkv", "datt", "dq", "dffh",
                     "logits", "logp", "pre_ids", "psi", "psi_eos", "cand_score", "cand_tok", "cand_ctc",
                     "sel"):
            setattr(s, name, getattr(sb, name).data_ptr())
        s.embed, s.pe = w.embed.data_ptr(), w.pe.data_ptr()
        s.dec_norm_g, s.dec_norm_b = w.dec_norm_g.data_ptr(), w.dec_norm_b.data_ptr()
        s.out_w, s.out_b = w.out_w.data_ptr(), w.out_b.data_ptr()
        s.layers = C.cast(layers, C.c_void_p).value
        s.rowmap, s.n_rows = sb.rowmap.data_ptr(), sb.S * sb.W
        s.out_w_q = w.out_w_q.data_ptr() if getattr(w, "out_w_q", None) is not None else None
        s.kv_half = 1 if sb.ckv.dtype == torch.float16 else 0
        s.kv_rows, s.kvflags = int(sb.kv_rows), sb.kvflags.data_ptr()
        # fp16 decoder mode (weights.PackedWeights(dec_dtype="float16")): needs the fp16 K|V caches
        if s.kv_half and getattr(w, "out_w_qh", None) is not None and all("wqkv_pph" in lw for lw in w.dec):
            # layer projections | partial products (| 4: output layer, measured and rejected: csrc/streams.hip).  SC_ACT_HALF is a
            # test hook like every SC_* switch: honoured only under SC_TEST_HOOKS=1, exactly as the C++ engine reads it (sc_hook)
            hooks = os.environ.get("SC_TEST_HOOKS", "0") not in ("", "0")
            s.out_w_qh, s.act_half = w.out_w_qh.data_ptr(), (int(os.environ.get("SC_ACT_HALF", "3")) & 7) if hooks else 3
        else:
            for i in range(len(w.dec)):
                layers[i].wqkv_pph = layers[i].wq_pph = layers[i].wo_pph = layers[i].wo2_pph = None
        if getattr(sb, "ctcxT", None) is not None:  # column-major CTC table copy
            s.ctcxT, s.tct = sb.ctcxT.data_ptr(), sb.ctcxT.shape[-1]
        if getattr(sb, "ph1", None) is not None:   # head-parallel decoder layers (include/scasr.h)
            s.ph1, s.ph2, s.ffn_part = sb.ph1.data_ptr(), sb.ph2.data_ptr(), sb.ffn_part.data_ptr()
            s.max_ffn_part = sb.ffn_part.shape[0]
        sb._sc_search_struct = (s, layers)
        return s

    def _sb_call(self, fn, sb, *extra):
        s = self.search_struct(sb)
        s.n_rows = int(getattr(sb, "n_rows_step", sb.S * sb.W))   # compaction bucket of this step
        self._chk(getattr(self.lib, fn)(C.addressof(s), *extra, self._stream()), fn)

    def ctc_extend_state(self, sb):
        self._sb_call("sc_ctc_extend_state", sb)

    def dec_embed(self, sb):
        self._sb_call("sc_dec_embed", sb)

    def dec_self_attn(self, sb, li):
        self._sb_call("sc_dec_self_attn", sb, li)

    def dec_cross_attn(self, sb, li):
        self._sb_call("sc_dec_cross_attn", sb, li)

    def decoder_layers(self, sb):
        self._sb_call("sc_decoder_layers", sb)

    # head-parallel decoder layers: 3 launches per layer (csrc/decoder_layer.hip)
    def dec_layer_self(self, sb, li, xin, xout, npart):
        self._sb_call("sc_dec_layer_self", sb, li, _p(xin), _p(xout), _p(sb.ffn_part), int(npart))

    def dec_layer_cross(self, sb, li, xin, xout):
        self._sb_call("sc_dec_layer_cross", sb, li, _p(xin), _p(xout))

    def dec_layer_ffn(self, sb, li, xin, xout):
        n = C.c_int(0)
        self._sb_call("sc_dec_layer_ffn", sb, li, _p(xin), _p(xout), _p(sb.ffn_part), int(sb.ffn_part.shape[0]),
                      C.byref(n))
        return int(n.value)

    # stream-resident decoder layers: 2 launches per layer (csrc/decoder_stream.hip)
    def dec_layer_stream(self, sb, li, xin, xout, xn_out, npart):
        self._sb_call("sc_dec_layer_stream", sb, li, _p(xin), _p(xout), _p(xn_out), _p(sb.ffn_part), int(npart))

    def dec_layer_ffn_xn(self, sb, li, xn):
        n = C.c_int(0)
        self._sb_call("sc_dec_layer_ffn_xn", sb, li, _p(xn), _p(sb.ffn_part), int(sb.ffn_part.shape[0]), C.byref(n))
        return int(n.value)

    def dec_output_logits(self, sb, xin, xout, npart):
        self._sb_call("sc_dec_output_logits", sb, _p(xin), _p(xout), _p(sb.ffn_part), int(npart))

    def logsoftmax_topk(self, sb):
        self._sb_call("sc_logsoftmax_topk", sb)

    def ctc_prefix_scan(self, sb, split_min=0):
        """split_min > 0: streams with that many frames to walk take the T-parallel kernel (same scores)"""
        self._sb_call("sc_ctc_prefix_scan_split", sb, int(split_min))

    def fuse_topw(self, sb):
        self._sb_call("sc_fuse_topw", sb)

    def beam_prune(self, sb):
        self._sb_call("sc_beam_prune", sb)

    def ctc_gather_state(self, sb, split_min=0):
        """split_min: the value the step's ctc_prefix_scan ran with (streams split over T leave segment states, not checkpoints)"""
        self._sb_call("sc_ctc_gather_state_split", sb, int(split_min))

    def decode_step(self, sb):
        """One beam-search step; replayed from a hipGraph after the first call
        (all launch geometry depends only on S, W, V, TCAP, LCAP; the per-step
        state is read from the device-side ctrl rows)."""
        if not self.use_graphs:
            self._sb_call("sc_decode_step", sb)
            return
        graphs = getattr(sb, "_sc_decode_graphs", None)
        if graphs is None:
            graphs = sb._sc_decode_graphs = {}
        g = graphs.get(int(sb.n_rows_step))     # one graph per compaction bucket
        st = self._stream()
        if g is None:
            if st == 0:
                raise _abi.ScasrError("hipGraph capture needs a non-default stream (StreamBatch.stream)")
            self._sb_call("sc_decode_step", sb)          # warm-up launch (also validates arguments)
            self._chk(self.lib.sc_graph_capture_begin(st), "sc_graph_capture_begin")
            try:
                self._sb_call("sc_decode_step", sb)
            finally:
                out = C.c_void_p()
                rc = self.lib.sc_graph_capture_end(st, C.byref(out))
            self._chk(rc, "sc_graph_capture_end")
            graphs[int(sb.n_rows_step)] = out
            return self._redo_after_capture(sb)
        self._chk(self.lib.sc_graph_launch(g, st), "sc_graph_launch")

    def prepare_decode(self, sb):
        """Capture the decode-step graph of every compaction bucket up front
        (graph instantiation costs ~0.1 s per bucket; without this it would land
        in the middle of the stream the first time a bucket size occurs).  Runs
        dry steps: the device ctrl rows are all-inactive, so every per-stream
        kernel exits and the dense kernels only touch scratch activations."""
        if not self.use_graphs or self._stream() == 0:
            return
        keep = sb.n_rows_step
        for nb in range(sb.row_bucket, sb.S + sb.row_bucket, sb.row_bucket):
            sb.n_rows_step = min(nb, sb.S) * sb.W
            if int(sb.n_rows_step) not in getattr(sb, "_sc_decode_graphs", {}):
                self.decode_step(sb)
        sb.n_rows_step = keep

    def _redo_after_capture(self, sb):
        # the warm-up launch before capture already executed this step once;
        # kernels only write the 1-cur side, so the captured graph is NOT launched again here.
        return None
