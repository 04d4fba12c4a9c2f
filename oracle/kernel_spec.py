"""ORACLE (test infrastructure, NOT product code): per-kernel specifications.

One torch-CPU function per C-ABI entry point of libscasr (include/scasr.h),
with the same argument meaning and the same buffer layouts, so that

* tests can run the product's host logic (speechcatcher_amd.engine) on CPU
  against the reference-port oracle (oracle/ref_port.py) without a GPU, and
* every HIP kernel is checked op-by-op against its spec on the GPU
  (tests/test_gpu_ops.py).

The arithmetic follows the reference (citations on each op, paths relative to
/root/reference); the decomposition (K/V caches, ancestor tables, ping-pong
hypothesis buffers) follows DESIGN.md.  Only tests/ may import this module.
"""
import math

import torch
import torch.nn.functional as F

LOGZERO = -10000000000.0


def _rows(t2d, rows):
    """Gather rows (index -1 -> zero row)."""
    rows = rows.to(torch.long)
    out = t2d[rows.clamp(min=0)]
    out = out.clone()
    out[rows < 0] = 0
    return out


class SpecBackend:
    fused_layers = True    # decode_step(): head-parallel layer ops when the batch has their buffers

    name = "spec"

    # ------------------------------------------------------------------
    # frontend: speechcatcher/model/frontend/stft_frontend.py:87-154 +
    # MVN / trimming of speechcatcher/speech2text_streaming.py:355-389
    # ------------------------------------------------------------------
    def logmel(self, w, pcm, pcap, jobs, n_jobs, max_keep, featbuf):
        cfg = w.cfg
        jobs = jobs.cpu().numpy()
        for j in range(n_jobs):
            s, seg_start, seg_len, eff_len, lo, n, dst0, _ = [int(v) for v in jobs[j]]
            x = torch.zeros(eff_len)
            x[:seg_len] = pcm[s, seg_start:seg_start + seg_len]
            st = torch.stft(x.unsqueeze(0), n_fft=cfg.n_fft, hop_length=cfg.hop_length,
                            win_length=cfg.win_length, window=w.window, center=True,
                            normalized=False, onesided=True, return_complex=True).transpose(1, 2)
            power = st.real ** 2 + st.imag ** 2
            mel = torch.clamp(torch.matmul(power, w.mel_fb), min=1e-10).log()[0]
            if w.has_mvn:
                if w.mvn_is_f64:
                    mel = ((mel.to(torch.float64) - w.mean64) / w.std64).to(torch.float32)
                else:
                    mel = (mel - w.mean64.to(torch.float32)) / w.std64.to(torch.float32)
            featbuf[dst0:dst0 + n] = mel[lo:lo + n]

    # ------------------------------------------------------------------
    # Conv2dSubsampling: speechcatcher/model/encoder/subsampling.py:87-98
    # ------------------------------------------------------------------
    def conv1(self, w, featbuf, jobs, n_jobs, max_t1, c1):
        cfg = w.cfg
        d, F1 = cfg.d_model, cfg.conv_freq1
        jobs = jobs.cpu().numpy()
        wt = w.conv1_w.view(d, 1, 3, 3)
        c1v = c1.view(-1, F1, d)
        for j in range(n_jobs):
            src0, t_in, r0, t1 = [int(v) for v in jobs[j]]
            x = featbuf[src0:src0 + t_in].view(1, 1, t_in, cfg.n_mels)
            y = torch.relu(F.conv2d(x, wt, w.conv1_b, stride=2))[0]  # (d, t1, F1)
            c1v[r0:r0 + t1] = y.permute(1, 2, 0)

    def gemm(self, A, a_rows, lda, W, bias, C, c_rows, ldc, M, N, K, relu=False,
             conv_f1=0, residual=False):
        """C[c_rows[m], :N] (+)= A[a_rows[m], :] . W^T + bias.

        A is addressed as a flat buffer: element (m, k) lives at
        a_rows[m]*lda + kofs(k) where kofs(k) = k, or for the implicit conv
        (conv_f1 > 0, lda = channels): tap = k // lda, kofs = ((tap//3)*conv_f1
        + tap%3)*lda + k%lda.  ``residual`` adds the previous content of C."""
        Af = A.reshape(-1)
        if a_rows is None:
            base = torch.arange(M, dtype=torch.long) * lda
        else:
            base = a_rows.to(torch.long)[:M] * lda
        k = torch.arange(K, dtype=torch.long)
        if conv_f1 > 0:
            tap = k // lda
            kofs = ((tap // 3) * conv_f1 + tap % 3) * lda + k % lda
        else:
            kofs = k
        a = Af[base.unsqueeze(1) + kofs.unsqueeze(0)]
        out = F.linear(a, W[:N, :K], bias)
        if relu:
            out = torch.relu(out)
        Cf = C.reshape(-1)
        if c_rows is None:
            cb = torch.arange(M, dtype=torch.long) * ldc
        else:
            cb = c_rows.to(torch.long)[:M] * ldc
        idx = cb.unsqueeze(1) + torch.arange(N, dtype=torch.long).unsqueeze(0)
        if residual:
            out = Cf[idx] + out
        Cf[idx] = out

    def copy_rows(self, src, src_rows, dst, dst_rows, n, width):
        dst[dst_rows.to(torch.long)[:n]] = src[src_rows.to(torch.long)[:n]].clone()

    def layernorm(self, src, src_rows, dst, dst_rows, M, g, b, eps=1e-12):
        """speechcatcher/model/layers/normalization.py:7-24"""
        x = src[:M] if src_rows is None else _rows(src, src_rows[:M])
        y = F.layer_norm(x, (x.size(-1),), g, b, eps)
        if dst_rows is None:
            dst[:M] = y
        else:
            dst[dst_rows.to(torch.long)[:M]] = y

    def log_softmax_rows(self, x, rows, n, V):
        r = rows.to(torch.long)[:n]
        x[r] = torch.log_softmax(x[r], dim=-1)

    # ------------------------------------------------------------------
    # block assembly: contextual_block_transformer_encoder.py:354-380
    # ------------------------------------------------------------------
    def block_pack(self, w, subbuf, jobs, nb, R, xblk):
        cfg = w.cfg
        d = cfg.d_model
        jobs = jobs.cpu().numpy()
        x = xblk[:nb * R].view(nb, R, d)
        sq = math.sqrt(d)
        for b in range(nb):
            src0, clen, pe_f, pe_c, short, _ = [int(v) for v in jobs[b]]
            chunk = subbuf[src0:src0 + clen]
            if short:
                x[b, :clen] = chunk * sq + w.pe[pe_f:pe_f + clen]
                continue
            x[b] = 0
            x[b, 1:clen + 1] = chunk * sq + w.pe[pe_f:pe_f + clen]
            x[b, R - 1] = chunk.mean(0) * sq + w.pe[pe_c]

    def ctx_handoff(self, x, R, jobs, ns, state, layer):
        """contextual_block_encoder_layer.py:253-269 (also the prev_addin
        chain of contextual_block_transformer_encoder.py:370-380)."""
        d = x.size(-1)
        jobs = jobs.cpu().numpy()
        for j in range(ns):
            b0, nbk, srow, has = [int(v) for v in jobs[j]]
            xv = x[b0 * R:(b0 + nbk) * R].view(nbk, R, d)
            v = state[srow + layer].clone() if has else xv[0, R - 1].clone()
            last = xv[:, R - 1].clone()
            xv[0, 0] = v
            if nbk > 1:
                xv[1:, 0] = last[:-1]
            state[srow + layer] = last[-1]

    def enc_attention(self, qkv, att, nblk, R, H, masked):
        """multi_head_attention.py:92-133 on (nblk, R, d) blocks.  masked:
        query row 0 fully masked (-> zeros), key column R-1 masked
        (contextual_block_transformer_encoder.py:524-528)."""
        d = att.size(-1)
        dk = d // H
        q, k, v = qkv[:nblk * R].view(nblk, R, 3, H, dk).unbind(2)
        q, k, v = (t.transpose(1, 2) for t in (q, k, v))
        scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
        if masked:
            mask = torch.zeros(R, R)
            mask[1:, :R - 1] = 1
            mask = mask.view(1, 1, R, R)
            scores = scores.masked_fill(mask == 0, torch.finfo(scores.dtype).min)
            p = torch.softmax(scores, dim=-1).masked_fill(mask == 0, 0.0)
        else:
            p = torch.softmax(scores, dim=-1)
        o = torch.matmul(p, v).transpose(1, 2).reshape(nblk * R, d)
        att[:nblk * R] = o

    def glu_dwconv_bn_swish(self, y, B, T, Cc, ksize, dw_w, dw_b, bn_g, bn_b, bn_mean, bn_var, eps, out):
        """convolution.py:103-112 on channels-last data."""
        yv = y.view(B, T, 2 * Cc).transpose(1, 2)
        a, g = yv[:, :Cc], yv[:, Cc:]
        z = a * torch.sigmoid(g)
        z = F.conv1d(z, dw_w.view(Cc, 1, ksize), dw_b, padding=(ksize - 1) // 2, groups=Cc)
        z = F.batch_norm(z, bn_mean, bn_var, bn_g, bn_b, False, 0.0, eps)
        z = z * torch.sigmoid(z)
        out.view(B, T, Cc).copy_(z.transpose(1, 2))

    def relpos_attention(self, qkv, p, bias_u, bias_v, out, B, T, H, mask=None):
        """multi_head_attention.py:343-375 incl. rel_shift :300-314 and the masks of :366-372
        (mask: uint8 [B][T] over keys or [B][T][T]; 0 = masked out)."""
        d = out.shape[-1]
        dk = d // H
        q, k, v = qkv.view(B, T, 3, H, dk).unbind(2)
        q, k, v = (t.transpose(1, 2) for t in (q, k, v))
        pp = p.view(1, T, H, dk).transpose(1, 2)
        ac = torch.matmul(q + bias_u.view(1, H, 1, dk), k.transpose(-2, -1))
        bd = torch.matmul(q + bias_v.view(1, H, 1, dk), pp.transpose(-2, -1))
        zp = torch.zeros((B, H, T, 1))
        bd = torch.cat([zp, bd], dim=-1).view(B, H, T + 1, T)[:, :, 1:].reshape(B, H, T, T)
        scores = (ac + bd) / math.sqrt(dk)
        if mask is not None:
            mm = (mask.view(B, 1, 1, T) if mask.dim() == 2 else mask.view(B, 1, T, T)) == 0
            att = torch.softmax(scores.masked_fill(mm, torch.finfo(scores.dtype).min), dim=-1).masked_fill(mm, 0.0)
        else:
            att = torch.softmax(scores, dim=-1)
        out.view(B, T, d).copy_(torch.matmul(att, v).transpose(1, 2).reshape(B, T, d))

    def encoder_layers(self, w, x, nblk, R, masked, jobs, ns, past_ctx, xn, qkv, att, ffh):
        """contextual_block_encoder_layer.py:215-271 x n_layers"""
        from speechcatcher_amd.weights import ffn_fused_supported
        cfg = w.cfg
        d, Fd, M = cfg.d_model, cfg.ffn_dim, nblk * R
        for li, lw in enumerate(w.enc):
            self.layernorm(x, None, xn, None, M, lw["ln1_g"], lw["ln1_b"])
            self.gemm(xn, None, d, lw["wqkv"], lw["bqkv"], qkv, None, 3 * d, M, 3 * d, d)
            self.enc_attention(qkv, att, nblk, R, cfg.enc_heads, masked)
            self.gemm(att, None, d, lw["wo"], lw["bo"], x, None, d, M, d, d, residual=True)
            self.layernorm(x, None, xn, None, M, lw["ln2_g"], lw["ln2_b"])
            if ffn_fused_supported(d, Fd):
                self.ffn_ln(xn, None, M, d, Fd, lw["w1_p"], lw["b1"], lw["w2_p"], lw["b2"], x, None, None, None)
            else:
                self.gemm(xn, None, d, lw["w1"], lw["b1"], ffh, None, Fd, M, Fd, d, relu=True)
                self.gemm(ffh, None, Fd, lw["w2"], lw["b2"], x, None, d, M, d, Fd, residual=True)
            if masked:
                self.ctx_handoff(x, R, jobs, ns, past_ctx, li)

    # ------------------------------------------------------------------
    # search-side ops.  ``sb`` is the StreamBatch (buffers + ctrl).
    # ------------------------------------------------------------------
    def ctc_extend_state(self, sb):
        """ctc_prefix_score_full.py:349-368"""
        ctrl = sb.ctrl.cpu().numpy()
        V = sb.cfg.vocab_size
        for s in range(sb.S):
            act, cur, fin, T, L, nh, has, told = [int(v) for v in ctrl[s]]
            if not act or told >= T:
                continue
            if getattr(sb, "ctcxT", None) is not None:   # column-major copy of the new table rows (sc_ctc_extend_state)
                sb.ctcxT.view(sb.S, V, -1)[s, :, told:T] = sb.ctcx.view(sb.S, sb.TCAP, V)[s, told:T].t()
            if not has:
                continue
            xb = sb.ctcx.view(sb.S, sb.TCAP, V)[s, :, sb.cfg.blank_id]
            r = sb.ctc_r[cur, s]
            for t in range(max(told, 1), T):
                r[t, 0, :nh] = LOGZERO
                r[t, 1, :nh] = r[t - 1, 1, :nh] + xb[t]
                if getattr(sb, "ctc_rs", None) is not None:      # log(exp r^n + exp r^b): r^n is logzero here
                    sb.ctc_rs[cur, s, t, :nh] = r[t, 1, :nh]

    def dec_embed(self, sb):
        """transformer_decoder.py:231 + positional_encoding.py:64-74"""
        w, cfg = sb.w, sb.cfg
        ctrl = sb.ctrl.cpu().numpy()
        sq = math.sqrt(cfg.d_model)
        for s in range(sb.S):
            act, cur, fin, T, L, nh, has, _ = [int(v) for v in ctrl[s]]
            if not act:
                continue
            tok = sb.yseq[cur, s, :nh, L - 1].to(torch.long)
            sb.dx[s * sb.W:s * sb.W + nh] = w.embed[tok] * sq + w.pe[L - 1]

    def dec_self_attn(self, sb, li):
        """decoder_layer.py:85-101 with a true K/V cache: the new K/V row goes
        into the pool row beam_prune gave the hypothesis (anc[L-1][h]); older rows
        are found through the ancestor table.  (A14: identical to re-projecting
        the output cache.)"""
        cfg = sb.cfg
        d, H, W = cfg.d_model, cfg.dec_heads, sb.W
        dk = d // H
        ctrl = sb.ctrl.cpu().numpy()
        skv = sb.skv.view(sb.S, cfg.dec_layers, int(sb.kv_rows), 2 * d)
        for s in range(sb.S):
            act, cur, fin, T, L, nh, has, _ = [int(v) for v in ctrl[s]]
            if not act:
                continue
            rows = slice(s * W, s * W + nh)
            qkv = sb.dqkv[rows]
            anc = sb.anc[cur, s, :L, :nh].to(torch.long)           # (L, nh) pool rows
            skv[s, li, anc[L - 1]] = qkv[:, d:].to(skv.dtype)
            kv = skv[s, li][anc].to(torch.float32)                  # (L, nh, 2d)
            k = kv[..., :d].permute(1, 0, 2).reshape(nh, L, H, dk).transpose(1, 2)
            v = kv[..., d:].permute(1, 0, 2).reshape(nh, L, H, dk).transpose(1, 2)
            q = qkv[:, :d].reshape(nh, 1, H, dk).transpose(1, 2)
            sc = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
            p = torch.softmax(sc, dim=-1)
            sb.datt[rows] = torch.matmul(p, v).transpose(1, 2).reshape(nh, d)

    def dec_cross_attn(self, sb, li):
        """decoder_layer.py:106-115, K/V projected once per encoder frame and
        shared by all hypotheses of the stream."""
        cfg = sb.cfg
        d, H, W = cfg.d_model, cfg.dec_heads, sb.W
        dk = d // H
        ctrl = sb.ctrl.cpu().numpy()
        ckv = sb.ckv.view(sb.S, cfg.dec_layers, sb.TCAP, 2 * d)
        for s in range(sb.S):
            act, cur, fin, T, L, nh, has, _ = [int(v) for v in ctrl[s]]
            if not act:
                continue
            rows = slice(s * W, s * W + nh)
            kv = ckv[s, li, :T]
            k = kv[:, :d].reshape(1, T, H, dk).transpose(1, 2)
            v = kv[:, d:].reshape(1, T, H, dk).transpose(1, 2)
            q = sb.dq[rows].reshape(nh, 1, H, dk).transpose(1, 2)
            sc = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
            p = torch.softmax(sc, dim=-1)
            sb.datt[rows] = torch.matmul(p, v).transpose(1, 2).reshape(nh, d)

    def gemm_ln(self, A, a_rows, lda, W, bias, C, c_rows, ldc, M, N, K, ln_g, ln_b, ln_out,
                relu=False, conv_f1=0, residual=False, eps=1e-12, ln_at_crows=False):
        """sc_gemm_ln: GEMM (+bias/ReLU/residual) then LayerNorm of the produced rows
        (ln_at_crows: ln_out rows follow c_rows, SC_GEMM_LN_AT_CROWS)."""
        self.gemm(A, a_rows, lda, W, bias, C, c_rows, ldc, M, N, K, relu=relu, conv_f1=conv_f1, residual=residual)
        self.layernorm(C, c_rows, ln_out, c_rows if ln_at_crows else None, M, ln_g, ln_b, eps)

    def proj_ln_proj(self, A, lda, W1, b1, X, ldx, ln_g, ln_b, XN, W2, b2, Q, M, D, eps=1e-12, rows=None):
        """sc_proj_ln_proj: X += A.W1^T + b1; XN = LN(X) (optional output);
        Q = LN(X).W2^T + b2 (optional).  W1 / W2 arrive in the lane order of
        sc_pack_lane_weight."""
        from speechcatcher_amd.weights import unpack_lane_weight
        xn = XN if XN is not None else torch.empty_like(X)
        self.gemm_ln(A, rows, lda, unpack_lane_weight(W1), b1, X, rows, ldx, M, D, D, ln_g, ln_b, xn,
                     residual=True, eps=eps, ln_at_crows=rows is not None)
        if W2 is not None:
            self.gemm(xn, rows, D, unpack_lane_weight(W2), b2, Q, rows, D, M, D, D)

    def ffn_ln(self, XN, rows, M, D, F, W1p, b1, W2p, b2, X, ln_g, ln_b, ln_out, eps=1e-12):
        """sc_ffn_ln: X[r] += W2.relu(W1.XN[r] + b1) + b2; ln_out[r] = LN(X[r]) (optional).
        W1p / W2p arrive in the fragment order of sc_pack_panel_weight."""
        from speechcatcher_amd.weights import unpack_panel_weight
        W1, W2 = unpack_panel_weight(W1p), unpack_panel_weight(W2p)
        nrow = XN.reshape(-1, D).shape[0]
        h = torch.empty(nrow, F, dtype=torch.float32)
        self.gemm(XN, rows, D, W1, b1, h, rows, F, M, F, D, relu=True)
        if ln_out is not None:
            self.gemm_ln(h, rows, F, W2, b2, X, rows, D, M, D, F, ln_g, ln_b, ln_out, residual=True, eps=eps,
                         ln_at_crows=rows is not None)
        else:
            self.gemm(h, rows, F, W2, b2, X, rows, D, M, D, F, residual=True)

    def rowtile_proj(self, A, M, D, Wp, bias, N, C_out, ln_g=None, ln_b=None, R=None, g2=None, b2=None, LN2=None,
                     eps=1e-12):
        """sc_rowtile_proj: C = [LN](A) . W^T + bias [+ R] [-> LN2]; W in sc_pack_panel_weight order
        (norm1 + q|k|v Linear, output Linear + residual + norm2 of contextual_block_encoder_layer.py:190-240)."""
        from speechcatcher_amd.weights import unpack_panel_weight
        a = A.reshape(-1, A.shape[-1])[:M, :D].to(torch.float32)
        if ln_g is not None:
            a = torch.nn.functional.layer_norm(a, (D,), ln_g, ln_b, eps)
        c = a @ unpack_panel_weight(Wp).t()
        if bias is not None:
            c = c + bias
        if R is not None:
            c = c + R.reshape(-1, R.shape[-1])[:M, :N]
        C_out.reshape(-1, C_out.shape[-1])[:M, :N] = c
        if LN2 is not None:
            LN2.reshape(-1, D)[:M] = torch.nn.functional.layer_norm(c, (D,), g2, b2, eps)

    def ffn_ln_proj(self, XN, rows, M, D, F, W1p, b1, W2p, b2, Xin, Xout, ln_g, ln_b, Wq, bq, Q, N, eps=1e-12):
        """sc_ffn_ln_proj: Xout[r] = Xin[r] + FFN(XN[r]); Q[r] = LN(Xout[r]) . Wq^T + bq.
        W1p / W2p in sc_pack_panel_weight order, Wq in sc_pack_lane_weight order."""
        from speechcatcher_amd.weights import unpack_lane_weight
        r = torch.arange(M) if rows is None else rows[:M].to(torch.long)
        Xout.reshape(-1, D)[r] = Xin.reshape(-1, D)[r]
        xn = torch.empty_like(Xout)
        self.ffn_ln(XN, rows, M, D, F, W1p, b1, W2p, b2, Xout, ln_g, ln_b, xn, eps=eps)
        self.gemm(xn, rows, D, unpack_lane_weight(Wq), bq, Q, rows, N, M, N, D)

    PANEL_DIMS = (64, 128, 256)   # sc_proj_ln_proj_supported

    def decoder_layers(self, sb, fuse_logits=False):
        """decoder_layer.py:60-132 x n_layers, the launch sequence of sc_decoder_layers /
        sc_decode_step: every LayerNorm but the first is fused into the kernel producing its
        input; dense ops run over the compacted rows sb.rowmap[:n_rows_step] of the active
        streams.  In the chained form the FFN's reduce kernel also projects the next layer's
        Q|K|V (or the output layer: returns True, logits written), x ping-pongs dx <-> dxn and
        the LayerNorm before the FFN lives in dq.  Otherwise leaves after_norm(x) in dxn."""
        from speechcatcher_amd.weights import ffn_fused_supported
        w, cfg = sb.w, sb.cfg
        d, Fd, V = cfg.d_model, cfg.ffn_dim, cfg.vocab_size
        n = int(sb.n_rows_step)
        rows = sb.rowmap[:n]
        panel = d in self.PANEL_DIMS
        fused = ffn_fused_supported(d, Fd)
        chain = panel and fused
        x, xalt = sb.dx, sb.dxn
        ffn_in = sb.dq if chain else sb.dxn
        ln1 = sb.dq if chain else sb.dxn
        self.layernorm(sb.dx, rows, ln1, rows, n, w.dec[0]["ln1_g"], w.dec[0]["ln1_b"])
        self.gemm(ln1, rows, d, w.dec[0]["wqkv"], w.dec[0]["bqkv"], sb.dqkv, rows, 3 * d, n, 3 * d, d)
        for li, lw in enumerate(w.dec):
            last = li + 1 == len(w.dec)
            ng = w.dec_norm_g if last else w.dec[li + 1]["ln1_g"]
            nb = w.dec_norm_b if last else w.dec[li + 1]["ln1_b"]
            self.dec_self_attn(sb, li)
            if panel:
                self.proj_ln_proj(sb.datt, d, lw["wo_p"], lw["bo"], x, d, lw["ln2_g"], lw["ln2_b"], None,
                                  lw["wq_p"], lw["bq"], sb.dq, n, d, rows=rows)
                self.dec_cross_attn(sb, li)
                self.proj_ln_proj(sb.datt, d, lw["wo2_p"], lw["bo2"], x, d, lw["ln3_g"], lw["ln3_b"], ffn_in,
                                  None, None, None, n, d, rows=rows)
            else:
                self.gemm_ln(sb.datt, rows, d, lw["wo"], lw["bo"], x, rows, d, n, d, d,
                             lw["ln2_g"], lw["ln2_b"], sb.dxn, residual=True, ln_at_crows=True)
                self.gemm(sb.dxn, rows, d, lw["wq"], lw["bq"], sb.dq, rows, d, n, d, d)
                self.dec_cross_attn(sb, li)
                self.gemm_ln(sb.datt, rows, d, lw["wo2"], lw["bo2"], x, rows, d, n, d, d,
                             lw["ln3_g"], lw["ln3_b"], sb.dxn, residual=True, ln_at_crows=True)
            if chain and (not last or (fuse_logits and getattr(w, "out_w_q", None) is not None)):
                if not last:
                    nx = w.dec[li + 1]
                    self.ffn_ln_proj(ffn_in, rows, n, d, Fd, lw["w1_p"], lw["b1"], lw["w2_p"], lw["b2"], x, xalt, ng, nb,
                                     nx["wqkv_q"], nx["bqkv"], sb.dqkv, 3 * d)
                else:
                    self.ffn_ln_proj(ffn_in, rows, n, d, Fd, lw["w1_p"], lw["b1"], lw["w2_p"], lw["b2"], x, xalt, ng, nb,
                                     w.out_w_q, w.out_b, sb.logits, V)
                    return True
                x, xalt = xalt, x
                continue
            ln_next = (sb.dxn if x is sb.dx else sb.dx) if chain else sb.dxn
            if fused:
                self.ffn_ln(ffn_in, rows, n, d, Fd, lw["w1_p"], lw["b1"], lw["w2_p"], lw["b2"], x, ng, nb, ln_next)
            else:
                self.gemm(ffn_in, rows, d, lw["w1"], lw["b1"], sb.dffh, rows, Fd, n, Fd, d, relu=True)
                self.gemm_ln(sb.dffh, rows, Fd, lw["w2"], lw["b2"], x, rows, d, n, d, Fd, ng, nb, ln_next,
                             residual=True, ln_at_crows=True)
            if not last:
                nx = w.dec[li + 1]
                self.gemm(ln_next, rows, d, nx["wqkv"], nx["bqkv"], sb.dqkv, rows, 3 * d, n, 3 * d, d)
            elif ln_next is not sb.dxn:
                self.copy_rows(ln_next, rows, sb.dxn, rows, n, d)
        return False

    # ------------------------------------------------------------------
    # head-parallel decoder layers (csrc/decoder_layer.hip, include/scasr.h sc_dec_layer_*): the same
    # arithmetic as decoder_layers() above, cut into three ops per layer.  x ping-pongs xin -> xout.
    # ------------------------------------------------------------------
    def _active_rows(self, sb):
        ctrl = sb.ctrl.cpu().numpy()
        for s in range(sb.S):
            act, cur, fin, T, L, nh, has, _ = [int(v) for v in ctrl[s]]
            if act and nh > 0:
                yield s, cur, T, L, nh

    # heads per workgroup of the HIP kernels (csrc/decoder_layer.hip: HPW): the partial products of `dec_hpw`
    # consecutive heads are summed before they are stored, a row carries H / dec_hpw of them
    dec_hpw = 1

    def _head_partials(self, ctx, wo, H, nh):
        """ph[w, g, :] = sum over the heads h of group g (in head order) of ctx[w, h*dk:(h+1)*dk] . wo[:, h*dk:(h+1)*dk]^T;
        rows >= nh are zero"""
        Wn, d = ctx.shape
        dk = d // H
        hpw = self.dec_hpw
        ph = torch.zeros(Wn, H // hpw, d)
        for h in range(H):
            part = ctx[:nh, h * dk:(h + 1) * dk] @ wo[:, h * dk:(h + 1) * dk].t()
            ph[:nh, h // hpw] = part if h % hpw == 0 else ph[:nh, h // hpw] + part
        return ph

    def _ph_view(self, buf, H):
        """the partial-product buffer [S*W][H][d] as the kernels lay it out with H / dec_hpw partials per row"""
        n, _, d = buf.shape
        nph = H // self.dec_hpw
        return buf.view(-1)[:n * nph * d].view(n, nph, d)

    def dec_layer_self(self, sb, li, xin, xout, npart):
        """sc_dec_layer_self: x (embedding for layer 0, else residual + feed-forward partial sums + b2 of the
        layer before) -> xout; norm1, q|k|v, K|V append, self-attention (decoder_layer.py:85-101); ph1 = the
        per-head partial products of self_attn.linear_out."""
        w, cfg = sb.w, sb.cfg
        d, H, W = cfg.d_model, cfg.dec_heads, sb.W
        lw = w.dec[li]
        sq = math.sqrt(d)
        for s, cur, T, L, nh in self._active_rows(sb):
            rows = slice(s * W, (s + 1) * W)
            if li == 0:
                hyp = torch.clamp(torch.arange(W), max=nh - 1)
                tok = sb.yseq[cur, s, hyp, L - 1].to(torch.long)
                x = w.embed[tok] * sq + w.pe[L - 1]
            else:
                y = sb.ffn_part[0, rows].clone()
                for z in range(1, npart):
                    y = y + sb.ffn_part[z, rows]
                x = xin[rows] + (y + w.dec[li - 1]["b2"])
            xout[rows] = x
            xn = torch.nn.functional.layer_norm(x, (d,), lw["ln1_g"], lw["ln1_b"], cfg.ln_eps)
            sb.dqkv[rows] = xn @ lw["wqkv"].t() + lw["bqkv"]
        self.dec_self_attn(sb, li)     # dqkv -> K|V rows appended to skv, context of rows < nh in datt
        for s, cur, T, L, nh in self._active_rows(sb):
            rows = slice(s * W, (s + 1) * W)
            self._ph_view(sb.ph1, H)[rows] = self._head_partials(sb.datt[rows], lw["wo"], H, nh)

    def dec_layer_cross(self, sb, li, xin, xout):
        """sc_dec_layer_cross: x = xin + bo + sum_h ph1 -> xout; norm2, q, cross-attention
        (decoder_layer.py:101-115); ph2 = per-head partial products of src_attn.linear_out."""
        w, cfg = sb.w, sb.cfg
        d, H, W = cfg.d_model, cfg.dec_heads, sb.W
        lw = w.dec[li]
        for s, cur, T, L, nh in self._active_rows(sb):
            rows = slice(s * W, (s + 1) * W)
            ph1 = self._ph_view(sb.ph1, H)
            y = ph1[rows, 0].clone()
            for h in range(1, ph1.shape[1]):
                y = y + ph1[rows, h]
            x = xin[rows] + (y + lw["bo"])
            xout[rows] = x
            xn = torch.nn.functional.layer_norm(x, (d,), lw["ln2_g"], lw["ln2_b"], cfg.ln_eps)
            sb.dq[rows] = xn @ lw["wq"].t() + lw["bq"]
        self.dec_cross_attn(sb, li)
        for s, cur, T, L, nh in self._active_rows(sb):
            rows = slice(s * W, (s + 1) * W)
            self._ph_view(sb.ph2, H)[rows] = self._head_partials(sb.datt[rows], lw["wo2"], H, nh)

    def dec_layer_ffn(self, sb, li, xin, xout):
        """sc_dec_layer_ffn over the compacted rows: x = xin + bo2 + sum_h ph2 -> xout; feed-forward of norm3(x)
        (decoder_layer.py:117-123) as partial sums - the spec writes ONE partial sum (returns 1)."""
        w, cfg = sb.w, sb.cfg
        d, H = cfg.d_model, cfg.dec_heads
        lw = w.dec[li]
        r = sb.rowmap[:int(sb.n_rows_step)].to(torch.long)
        ph2 = self._ph_view(sb.ph2, H)
        y = ph2[r, 0].clone()
        for h in range(1, ph2.shape[1]):
            y = y + ph2[r, h]
        x = xin[r] + (y + lw["bo2"])
        xout[r] = x
        xn = torch.nn.functional.layer_norm(x, (d,), lw["ln3_g"], lw["ln3_b"], cfg.ln_eps)
        hid = torch.relu(xn @ lw["w1"].t() + lw["b1"])
        sb.ffn_part[0, r] = hid @ lw["w2"].t()
        return 1

    # decode_step(): the stream-resident form of the layers (csrc/decoder_stream.hip, round 6) - 2 ops per layer
    stream_layers = False

    def dec_layer_stream(self, sb, li, xin, xout, xn_out, npart):
        """sc_dec_layer_stream: x (embedding for layer 0, else residual + feed-forward partial sums + b2 of the layer
        before); self-attention block (decoder_layer.py:85-101), cross-attention block (:106-115) -> xout;
        xn_out = norm3(xout) (:117).  xn_out may be sb.dq (the spec's own q scratch: overwritten last)."""
        w, cfg = sb.w, sb.cfg
        d, W = cfg.d_model, sb.W
        lw = w.dec[li]
        sq = math.sqrt(d)
        ln = torch.nn.functional.layer_norm
        for s, cur, T, L, nh in self._active_rows(sb):
            rows = slice(s * W, (s + 1) * W)
            if li == 0:
                hyp = torch.clamp(torch.arange(W), max=nh - 1)
                tok = sb.yseq[cur, s, hyp, L - 1].to(torch.long)
                x = w.embed[tok] * sq + w.pe[L - 1]
            else:
                y = sb.ffn_part[0, rows].clone()
                for z in range(1, npart):
                    y = y + sb.ffn_part[z, rows]
                x = xin[rows] + (y + w.dec[li - 1]["b2"])
            xout[rows] = x
            sb.dqkv[rows] = ln(x, (d,), lw["ln1_g"], lw["ln1_b"], cfg.ln_eps) @ lw["wqkv"].t() + lw["bqkv"]
        self.dec_self_attn(sb, li)     # dqkv -> K|V rows appended to skv, context of rows < nh in datt
        for s, cur, T, L, nh in self._active_rows(sb):
            rows = slice(s * W, (s + 1) * W)
            y = torch.zeros(W, d)
            y[:nh] = sb.datt[rows][:nh] @ lw["wo"].t()
            x = xout[rows] + (y + lw["bo"])
            xout[rows] = x
            sb.dq[rows] = ln(x, (d,), lw["ln2_g"], lw["ln2_b"], cfg.ln_eps) @ lw["wq"].t() + lw["bq"]
        self.dec_cross_attn(sb, li)
        for s, cur, T, L, nh in self._active_rows(sb):
            rows = slice(s * W, (s + 1) * W)
            y = torch.zeros(W, d)
            y[:nh] = sb.datt[rows][:nh] @ lw["wo2"].t()
            x = xout[rows] + (y + lw["bo2"])
            xout[rows] = x
            xn_out[rows] = ln(x, (d,), lw["ln3_g"], lw["ln3_b"], cfg.ln_eps)

    def dec_layer_ffn_xn(self, sb, li, xn):
        """sc_dec_layer_ffn_xn over the compacted rows: feed-forward of the rows xn (feed_forward.py:48-50) as partial
        sums by row id - the spec writes ONE partial sum (returns 1)."""
        lw = sb.w.dec[li]
        r = sb.rowmap[:int(sb.n_rows_step)].to(torch.long)
        hid = torch.relu(xn[r] @ lw["w1"].t() + lw["b1"])
        sb.ffn_part[0, r] = hid @ lw["w2"].t()
        return 1

    def dec_output_logits(self, sb, xin, xout, npart):
        """sc_dec_output_logits: residual + feed-forward partial sums + b2 -> xout; after_norm, output layer
        (transformer_decoder.py:243-249)."""
        w, cfg = sb.w, sb.cfg
        d = cfg.d_model
        r = sb.rowmap[:int(sb.n_rows_step)].to(torch.long)
        y = sb.ffn_part[0, r].clone()
        for z in range(1, npart):
            y = y + sb.ffn_part[z, r]
        x = xin[r] + (y + w.dec[-1]["b2"])
        xout[r] = x
        xn = torch.nn.functional.layer_norm(x, (d,), w.dec_norm_g, w.dec_norm_b, cfg.ln_eps)
        sb.logits[r] = xn @ w.out_w.t() + w.out_b

    def logsoftmax_topk(self, sb):
        """transformer_decoder.py:249 + pre-beam beam_search.py:150-154:
        logp = log_softmax(logits); ids = top-K of fl(w_dec * logp), descending,
        ties -> lowest index."""
        V, K, W = sb.cfg.vocab_size, sb.K, sb.W
        ctrl = sb.ctrl.cpu().numpy()
        wd = torch.tensor(sb.search.decoder_weight, dtype=torch.float32)
        for s in range(sb.S):
            act, cur, fin, T, L, nh, has, _ = [int(v) for v in ctrl[s]]
            if not act:
                continue
            rows = slice(s * W, s * W + nh)
            sb.logp[rows] = torch.log_softmax(sb.logits[rows], dim=-1)
            key = wd * sb.logp[rows]
            # stable descending sort == lowest index first among ties
            order = torch.sort(key, dim=-1, descending=True, stable=True).indices
            sb.pre_ids[rows] = order[:, :K].to(torch.int32)

    def ctc_prefix_scan(self, sb):
        """CTCPrefixScoreTH.__call__ (ctc_prefix_score_full.py:88-291) for the
        K pre-beam candidates of every live hypothesis.  Outputs
        psi[(s,h),k] = log_psi of candidate k (blank candidate -> logzero,
        eos candidate -> r_sum[T-1]), psi_eos[(s,h)] = r_sum[T-1],
        ctc_rnew[s,j,:,h*K+k] = r at the checkpoint frame 16 j + 15 (what the kernels store; the
        full-resolution r of the step stays in sb._rnew_full for ctc_gather_state)."""
        cfg = sb.cfg
        V, W, K = cfg.vocab_size, sb.W, sb.K
        ctrl = sb.ctrl.cpu().numpy()
        X = sb.ctcx.view(sb.S, sb.TCAP, V)
        for s in range(sb.S):
            act, cur, fin, T, L, nh, has, tctc = [int(v) for v in ctrl[s]]
            if not act:
                continue
            T = tctc if tctc > 0 else T      # scasr.h SC_C_TCTC: stale-table quirk after reset()
            x = X[s, :T]
            xb = x[:, cfg.blank_id]
            ids = sb.pre_ids[s * W:s * W + nh].to(torch.long)        # (nh, K)
            last = sb.yseq[cur, s, :nh, L - 1].to(torch.long)
            if has:
                r_prev = sb.ctc_r[cur, s, :T, :, :nh]                 # (T, 2, nh)
            else:
                r_prev = torch.full((T, 2, nh), LOGZERO)
                r_prev[:, 1] = torch.cumsum(xb, 0).unsqueeze(1)
            r_sum = torch.logsumexp(r_prev, 1)                       # (T, nh)
            xn = x[:, ids.reshape(-1)].view(T, nh, K)
            log_phi = r_sum.unsqueeze(2).repeat(1, 1, K)
            same = ids == last.unsqueeze(1)                          # (nh, K)
            log_phi = torch.where(same.unsqueeze(0), r_prev[:, 1].unsqueeze(2).expand(T, nh, K), log_phi)
            r = torch.full((T, 2, nh, K), LOGZERO)
            out_len = L - 1
            if out_len == 0:
                r[0, 0] = xn[0]
            start = min(max(out_len, 1), T)
            for t in range(start, T):
                rp = r[t - 1]
                r[t, 0] = torch.logsumexp(torch.stack([rp[0], log_phi[t - 1]]), 0) + xn[t]
                r[t, 1] = torch.logsumexp(torch.stack([rp[0], rp[1]]), 0) + xb[t]
            log_phi_x = torch.cat((log_phi[0].unsqueeze(0), log_phi[:-1]), 0) + xn
            psi = torch.logsumexp(torch.cat((log_phi_x[start:T], r[start - 1, 0].unsqueeze(0)), 0), 0)
            eos_val = r_sum[T - 1]                                   # (nh,)
            psi = torch.where(ids == cfg.eos_id, eos_val.unsqueeze(1).expand(nh, K), psi)
            psi = torch.where(ids == cfg.blank_id, torch.full_like(psi, LOGZERO), psi)
            sb.psi[s * W:s * W + nh] = psi
            sb.psi_eos[s * W:s * W + nh] = eos_val
            rf = r.reshape(T, 2, nh * K)
            if getattr(sb, "_rnew_full", None) is None:
                sb._rnew_full = {}
            sb._rnew_full[s] = rf
            nck = T // 16
            if nck:
                sb.ctc_rnew[s, :nck, :, :nh * K] = rf[15:16 * nck:16]

    def fuse_topw(self, sb):
        """beam_search.py:113-185 fusion + :723 per-hypothesis top-W.
        combined[v] = fl(fl(wd*logp[v]) + fl(wc*ctc[v])), ctc[v] = logpsi[v] - s_prev."""
        cfg = sb.cfg
        V, W, K = cfg.vocab_size, sb.W, sb.K
        ctrl = sb.ctrl.cpu().numpy()
        wd = torch.tensor(sb.search.decoder_weight, dtype=torch.float32)
        wc = torch.tensor(sb.search.ctc_weight, dtype=torch.float32)
        for s in range(sb.S):
            act, cur, fin, T, L, nh, has, _ = [int(v) for v in ctrl[s]]
            if not act:
                continue
            rows = slice(s * W, s * W + nh)
            ids = sb.pre_ids[rows].to(torch.long)
            if not float(wc) > 0.0:     # no CTC scorer (beam_search.py:925): decoder-only, "ctc" never enters the scores
                comb = wd * sb.logp[rows]
                order = torch.sort(comb, dim=-1, descending=True, stable=True).indices[:, :W]
                sb.cand_tok[rows] = order.to(torch.int32)
                sb.cand_score[rows] = comb.gather(1, order)
                sb.cand_ctc[rows] = 0.0
                continue
            lpsi = torch.full((nh, V), LOGZERO)
            lpsi.scatter_(1, ids, sb.psi[rows])
            lpsi[:, cfg.eos_id] = sb.psi_eos[rows]
            lpsi[:, cfg.blank_id] = LOGZERO
            s_prev = sb.ctc_s[cur, s, :nh].unsqueeze(1) if has else torch.zeros(nh, 1)
            ctc = lpsi - s_prev
            comb = wd * sb.logp[rows] + wc * ctc
            order = torch.sort(comb, dim=-1, descending=True, stable=True).indices[:, :W]
            sb.cand_tok[rows] = order.to(torch.int32)
            sb.cand_score[rows] = comb.gather(1, order)
            sb.cand_ctc[rows] = ctc.gather(1, order)

    def beam_prune(self, sb):
        """beam_search.py:721-809 + hypothesis.py:132-142: expand, stable
        descending sort on float64 totals, keep W, bookkeeping, stop flags."""
        cfg = sb.cfg
        W, K = sb.W, sb.K
        ctrl = sb.ctrl.cpu().numpy()
        flags = torch.zeros(sb.S, dtype=torch.int32)
        for s in range(sb.S):
            act, cur, fin, T, L, nh, has, _ = [int(v) for v in ctrl[s]]
            if not act:
                continue
            o = 1 - cur
            cands = []
            for h in range(nh):
                for j in range(W):
                    tot = float(sb.score[cur, s, h]) + float(sb.cand_score[s * W + h, j])
                    cands.append((tot, h, j))
            cands = sorted(cands, key=lambda c: c[0], reverse=True)[:W]
            any_eos = all_eos = best_eos = rep = False
            all_eos = True
            for i, (tot, h, j) in enumerate(cands):
                tok = int(sb.cand_tok[s * W + h, j])
                sb.yseq[o, s, i, :L] = sb.yseq[cur, s, h, :L]
                sb.yseq[o, s, i, L] = tok
                sb.xpos[o, s, i, :L] = sb.xpos[cur, s, h, :L]
                sb.xpos[o, s, i, L] = T - 1
                sb.score[o, s, i] = tot
                sb.sc_dec[o, s, i] = float(sb.sc_dec[cur, s, h]) + float(sb.logp[s * W + h, tok])
                sb.sc_ctc[o, s, i] = float(sb.sc_ctc[cur, s, h]) + float(sb.cand_ctc[s * W + h, j])
                sb.anc[o, s, :L, i] = sb.anc[cur, s, :L, h]   # pool rows of the parent's history incl. its newest token
                ids = sb.pre_ids[s * W + h].tolist()
                k = ids.index(tok) if tok in ids else -1
                if tok == cfg.blank_id:
                    sval = LOGZERO
                elif tok == cfg.eos_id:
                    sval = float(sb.psi_eos[s * W + h])
                elif k >= 0:
                    sval = float(sb.psi[s * W + h, k])
                else:
                    sval = LOGZERO
                sb.ctc_s[o, s, i] = sval
                sb.sel[s, i, 0] = h
                sb.sel[s, i, 1] = max(k, 0)
                is_eos = tok == cfg.eos_id
                any_eos |= is_eos
                all_eos &= is_eos
                if i == 0:
                    best_eos = is_eos
                if tok != cfg.sos_id and tok != cfg.eos_id:
                    if tok in sb.yseq[o, s, i, 1:L].tolist():
                        rep = True
            flags[s] = (1 if any_eos else 0) | (2 if best_eos else 0) | (4 if all_eos else 0) | (8 if rep else 0)
            # K|V pool rows of the new hypotheses' newest tokens (position L): the lowest rows none of them descends from
            nout = len(cands)
            if L + 1 <= sb.LCAP:
                NR = int(sb.kv_rows)
                used = set(sb.anc[o, s, :L, :nout].reshape(-1).tolist())
                free = [r for r in range(NR) if r not in used][:nout]
                sb.kvflags[s] = 1 if len(free) < nout else 0
                free += [NR - 1] * (nout - len(free))
                sb.anc[o, s, L, :nout] = torch.tensor(free, dtype=torch.int32)
        sb.flags.copy_(flags)

    def ctc_gather_state(self, sb):
        """CTCPrefixScorer.select_state (scorers.py:382-431)."""
        W, K = sb.W, sb.K
        ctrl = sb.ctrl.cpu().numpy()
        for s in range(sb.S):
            act, cur, fin, T, L, nh, has, tctc = [int(v) for v in ctrl[s]]
            if not act:
                continue
            T = tctc if tctc > 0 else T
            o = 1 - cur
            nout = min(W, nh * W)
            for i in range(nout):
                h, k = int(sb.sel[s, i, 0]), int(sb.sel[s, i, 1])
                sb.ctc_r[o, s, :T, :, i] = sb._rnew_full[s][:T, :, h * K + k]
                if getattr(sb, "ctc_rs", None) is not None:
                    sb.ctc_rs[o, s, :T, i] = torch.logsumexp(sb.ctc_r[o, s, :T, :, i], 1)

    def decode_step(self, sb):
        """One beam-search step for every active stream
        (beam_search.py:701-758)."""
        w, cfg = sb.w, sb.cfg
        n, d = int(sb.n_rows_step), cfg.d_model
        rows = sb.rowmap[:n]
        if getattr(sb, "ph1", None) is not None and self.fused_layers and self.stream_layers:
            # stream-resident layer kernels, 2 ops per layer (sc_decode_step: buckets of >= SC_STREAM_MIN_ROWS rows)
            xa, xb, npart = sb.dx, sb.dxn, 0
            for li in range(len(w.dec)):
                self.dec_layer_stream(sb, li, xa, xb, sb.dq, npart)
                npart = self.dec_layer_ffn_xn(sb, li, sb.dq)
                xa, xb = xb, xa
            self.dec_output_logits(sb, xa, xb, npart)
        elif getattr(sb, "ph1", None) is not None and self.fused_layers:
            # head-parallel layer kernels, 3 ops per layer (sc_decode_step takes this path for the same models)
            xa, xb, npart = sb.dx, sb.dxn, 0
            for li in range(len(w.dec)):
                self.dec_layer_self(sb, li, xa, xb, npart)
                self.dec_layer_cross(sb, li, xb, xa)
                npart = self.dec_layer_ffn(sb, li, xa, xb)
                xa, xb = xb, xa
            self.dec_output_logits(sb, xa, xb, npart)
        else:
            self.dec_embed(sb)
            if not self.decoder_layers(sb, fuse_logits=True):      # logits, or after_norm(x) in dxn
                self.gemm(sb.dxn, rows, d, w.out_w, w.out_b, sb.logits, rows, cfg.vocab_size, n, cfg.vocab_size, d)
        self.logsoftmax_topk(sb)
        use_ctc = sb.search.ctc_weight > 0
        if use_ctc:
            self.ctc_prefix_scan(sb)
        self.fuse_topw(sb)
        self.beam_prune(sb)
        if use_ctc:
            self.ctc_gather_state(sb)
