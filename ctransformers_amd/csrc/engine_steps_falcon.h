// falcon graph (llm_build_falcon, llama.cpp:2493-2798): prompt chunks and the token step — part of engine.cc (one translation unit: the HIP kernels are templates and file-local helpers of it); included there,
// inside namespace ctamd, after the launch helpers it uses.  Not a stand-alone header.

// llm_build_falcon (llama.cpp:2493-2798) for the nt tokens of a chunk: token_step_falcon's launches over rows of the chunk.
bool Engine::chunk_step_falcon(int c0, int nt, bool want_logits, std::string& err) {
    const int E = hp_.n_embd, G = hp_.n_embd_gqa(), F = hp_.n_ff, hd = hp_.head_dim();
    const int* d_pos = d_state_ + 1;
    if (l0_ == 0) {
        CT_LAUNCH(embed_row_kernel, dim3((unsigned)std::max(1, E / 256), (unsigned)nt), dim3(256), stream_, tok_embd_.raw, tok_embd_.type, E,
                  (const int*)d_tokens_, (const int*)d_state_, xb_);
    } else {
        HIP_OK(hipMemcpyAsync(xb_, xio_ + (size_t)c0 * E, (size_t)nt * E * 4, hipMemcpyDeviceToDevice, stream_));
    }
    MatvecArgs base = MatvecArgs();
    base.rope_cs = rope_cs_;
    base.pos = d_pos;
    base.n_ctx = n_ctx_;
    base.head_dim = hd;
    base.n_embd_gqa = G;
    base.v_stride = v_stride_;
    base.silu_tab = silu_tab_;
    base.gelu_tab = gelu_tab_;
    base.eps = hp_.rms_eps;
    for (int il = l0_; il < l1_; ++il) {
        const Layer& L = layers_[il];
        cur_layer_ = il;
        uint16_t* kc = kcache_ + (size_t)(il - l0_) * n_ctx_ * G;
        uint16_t* vc = vcache_ + (size_t)(il - l0_) * v_stride_ * G;
        {   // LayerNorm -> Q8_K -> fused QKV rows (f32, un-rotated), one row of E + 2G per token
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM;
            a.norm_w = L.attn_norm2 ? L.attn_norm2 : L.attn_norm;
            a.norm_b = L.attn_norm2 ? L.attn_norm2_b : L.attn_norm_b;
            a.out = qkv_tmp_b_;
            set_jobs(a, {{&L.wqkv, EPI_STORE}});
            if (!pf_matvec(a, xb_, E, nt, E + 2 * G, 0, "qkv", (double)L.wqkv.bytes, err)) return false;
        }
        if (site_on("rope_store"))
            CT_LAUNCH(falcon_rope_store_kernel, dim3((unsigned)(hp_.n_head + 2 * hp_.n_head_kv), (unsigned)nt), dim3((unsigned)(hd / 2)), stream_,
                      (const float*)qkv_tmp_b_, q_f16_b_, kc, vc, (const float*)rope_cs_, d_pos, hp_.n_head, hp_.n_head_kv, hd, n_ctx_,
                      v_stride_, falcon_fold_ ? 1 : 0);
        if (site_on("attn_fused")) launch_attention(kc, vc, nt);
        {   // Wo, kept apart: the residual is added after the MLP
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_PLAIN; a.out = attn_proj_b_;
            set_jobs(a, {{&L.wo, EPI_STORE}});
            if (!pf_matvec(a, attn_out_b_, E, nt, E, 0, "wo", (double)L.wo.bytes, err)) return false;
        }
        {   // LayerNorm(attn_norm) -> Q8_K -> W_up -> GELU (parallel block: the MLP reads the attention norm)
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.norm_w = L.attn_norm; a.norm_b = L.attn_norm_b; a.out = hb_;
            set_jobs(a, {{&L.w_up, EPI_GELU}});
            if (!pf_matvec(a, xb_, E, nt, F, 0, "ffn_up", (double)L.w_up.bytes, err)) return false;
        }
        {   // W_down -> (ffn + attn_out) + x
            MatvecArgs a = base;
            a.K = F; a.pro = PRO_PLAIN; a.out = xb_; a.res = attn_proj_b_; a.res2 = xb_;
            set_jobs(a, {{&L.w_down, EPI_ADD2}});
            if (!pf_matvec(a, hb_, F, nt, E, E, "down", (double)L.w_down.bytes, err)) return false;
        }
    }
    if (l1_ < hp_.n_layer) {
        HIP_OK(hipMemcpyAsync(xio_ + (size_t)c0 * E, xb_, (size_t)nt * E * 4, hipMemcpyDeviceToDevice, stream_));
    } else if (want_logits) {
        const float* xl = xb_ + (size_t)(nt - 1) * E;
        MatvecArgs a = base;
        a.K = E; a.pro = PRO_LAYERNORM; a.x = xl; a.norm_w = output_norm_; a.norm_b = output_norm_b_; a.out = d_logits_;
        set_jobs(a, {{&output_, EPI_STORE}});
        if (kq_can(a) && E <= 16384) { a.emb_out = d_emb_; set_head_fold(a, false); }   // as the llama head: final-norm output from the prologue, greedy pick in the epilogue
        else CT_LAUNCH((layernorm_f32_kernel<256>), dim3(1), dim3(256), stream_, xl, (const float*)output_norm_, (const float*)output_norm_b_,
                       d_emb_, E, hp_.rms_eps);
        if (!run_matvec(a, err)) return false;
        if (a.pick_ws) launch_pick(a);
    }
    CT_LAUNCH(advance_state_n_kernel, dim3(1), dim3(64), stream_, d_state_, nt);
    return true;
}

// llm_build_falcon (llama.cpp:2493-2798), one token: per layer
//   LayerNorm(attn_norm_2 or attn_norm) -> Q8_K -> fused Wqkv -> [neox RoPE, fp16 Q, KV append] -> attention -> Wo
//   LayerNorm(attn_norm) -> Q8_K -> W_up -> GELU table -> Q8_K -> W_down -> (ffn + attn_out) + x
bool Engine::token_step_falcon(bool want_logits, std::string& err) {
    const int E = hp_.n_embd, G = hp_.n_embd_gqa(), F = hp_.n_ff, hd = hp_.head_dim();
    const int* d_pos = d_state_ + 1;
    if (l0_ == 0 && cont_mode_) {
        // continuation step of a greedy chain (token_step, llama): the embedding row is in x_
    } else if (l0_ == 0) {
        if (site_on("embed")) {
            CT_LAUNCH(embed_row_kernel, dim3((unsigned)std::max(1, E / 256)), dim3(256), stream_, tok_embd_.raw, tok_embd_.type, E,
                      (const int*)d_tokens_, (const int*)d_state_, x_);
        }
    } else {  // inner pipeline stage: this token's residual-stream row was handed over by the previous stage
        CT_LAUNCH(stage_row_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), stream_, (const float*)xio_, x_, E,
                  (const int*)d_state_, 0);
    }
    MatvecArgs base = MatvecArgs();
    base.rope_cs = rope_cs_;
    base.pos = d_pos;
    base.n_ctx = n_ctx_;
    base.head_dim = hd;
    base.n_embd_gqa = G;
    base.v_stride = v_stride_;
    base.silu_tab = silu_tab_;
    base.gelu_tab = gelu_tab_;
    base.eps = hp_.rms_eps;
    base.dbg = env_int("CT_AMD_DBG", 0);
    base.dbg_sink = scores_; base.f16_tmp = f16_tmp_;
    bool bumped = false;
    for (int il = l0_; il < l1_; ++il) {
        const Layer& L = layers_[il];
        cur_layer_ = il;
        uint16_t* kc = kcache_ + (size_t)(il - l0_) * n_ctx_ * G;
        uint16_t* vc = vcache_ + (size_t)(il - l0_) * v_stride_ * G;
        bool fused_qa = false;
        {   // LayerNorm -> Q8_K -> fused QKV rows (f32, un-rotated)
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.x = x_;
            a.norm_w = L.attn_norm2 ? L.attn_norm2 : L.attn_norm;
            a.norm_b = L.attn_norm2 ? L.attn_norm2_b : L.attn_norm_b;
            a.out = qkv_tmp_;
            if (falcon_fold_) {   // NEOX RoPE, fp16 Q and the KV append in this launch's epilogue (rows reordered at load): no falcon_rope_store_kernel launch
                a.rope_neox = 1; a.q_f16 = q_f16_; a.kcache = kc; a.vcache = vc;
                set_jobs(a, {{&L.wq_v, EPI_ROPE_Q}, {&L.wk_v, EPI_ROPE_K}, {&L.wv_v, EPI_V}});
            } else {
                set_jobs(a, {{&L.wqkv, EPI_STORE}});
            }
            apply_trace(a, "qkv");
            fused_qa = falcon_fold_ && qa_can(L) && !only_site_ && !prof_ && (!trace_site_ || !strcmp(trace_site_, "qa"));
            if (fused_qa) {   // ... and the attention in the same launch (kernels_qa9.h, LayerNorm form)
                if (trace_site_) { a.dbg |= 32; a.dbg_sink = (float*)(trace_buf_ + 256); }
                if (!launch_qkv_attn(a, kc, vc, il, err)) return false;
            } else if (site_on("qkv")) {
                prof_begin("qkv", "matvec", (double)L.wqkv.bytes);
                if (!run_matvec(a, err)) return false;
                prof_end();
            }
        }
        if (!falcon_fold_ && site_on("rope_store")) {
            prof_begin("rope_store", "falcon_rope_store_kernel", 0.0);
            CT_LAUNCH(falcon_rope_store_kernel, dim3((unsigned)(hp_.n_head + 2 * hp_.n_head_kv)), dim3((unsigned)(hd / 2)), stream_,
                      (const float*)qkv_tmp_, q_f16_, kc, vc, (const float*)rope_cs_, d_pos, hp_.n_head, hp_.n_head_kv, hd, n_ctx_,
                      v_stride_, 0);
            prof_end();
        }
        if (!fused_qa && site_on("attn_fused")) {
            prof_begin("attn_fused", "attn_fused_exact_kernel", 0.0);
            launch_attention(kc, vc);
            prof_end();
        }
        {   // Q8_K(attn) -> Wo  (kept apart: the residual is added after the MLP)
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_PLAIN; a.x = attn_out_; a.out = attn_proj_;
            set_jobs(a, {{&L.wo, EPI_STORE}});
            apply_trace(a, "wo");
            if (site_on("wo")) {
                prof_begin("wo", "matvec", (double)L.wo.bytes);
                if (!run_matvec(a, err)) return false;
                prof_end();
            }
        }
        {   // LayerNorm(attn_norm) -> Q8_K -> W_up -> GELU   (the MLP reads the attention norm: parallel block)
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.x = x_; a.norm_w = L.attn_norm; a.norm_b = L.attn_norm_b; a.out = h_;
            set_jobs(a, {{&L.w_up, EPI_GELU}});
            apply_trace(a, "ffn_up");
            if (site_on("ffn_up")) {
                prof_begin("ffn_up", "matvec", (double)L.w_up.bytes);
                if (!run_matvec(a, err)) return false;
                prof_end();
            }
        }
        {   // Q8_K(h) -> W_down -> (ffn + attn_out) + x
            MatvecArgs a = base;
            a.K = F; a.pro = PRO_PLAIN; a.x = h_; a.out = x_; a.res = attn_proj_; a.res2 = x_;
            set_jobs(a, {{&L.w_down, EPI_ADD2}});
            if (il == hp_.n_layer - 1 && !only_site_ && !prof_ && kq_can(a)) { a.bump = d_state_; bumped = true; }   // as in token_step (llama)
            apply_trace(a, "down");
            if (site_on("down")) {
                prof_begin("down", "matvec", (double)L.w_down.bytes);
                if (!run_matvec(a, err)) return false;
                prof_end();
            }
        }
    }
    if (l1_ < hp_.n_layer) {  // hand this token's residual-stream row to the next stage
        CT_LAUNCH(stage_row_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), stream_, (const float*)x_, xio_, E,
                  (const int*)d_state_, 1);
    } else if (want_logits) {
        MatvecArgs a = base;
        a.K = E; a.pro = PRO_LAYERNORM; a.x = x_; a.norm_w = output_norm_; a.norm_b = output_norm_b_; a.out = d_logits_;
        set_jobs(a, {{&output_, EPI_STORE}});
        if (bumped) a.pos = nullptr;   // as in token_step (llama): the cursor was advanced already
        if (kq_can(a) && E <= 16384) { a.emb_out = d_emb_; set_head_fold(a, bumped); }   // as the llama head
        else if (!only_site_)
            CT_LAUNCH((layernorm_f32_kernel<256>), dim3(1), dim3(256), stream_, (const float*)x_, (const float*)output_norm_,
                      (const float*)output_norm_b_, d_emb_, E, hp_.rms_eps);
        apply_trace(a, "lm_head");
        if (site_on("lm_head")) {
            prof_begin("lm_head", "matvec", (double)output_.bytes);
            if (!run_matvec(a, err)) return false;
            prof_end();
        }
        if (a.pick_ws && !only_site_) launch_pick(a);
    }
    if (!only_site_ && !bumped) CT_LAUNCH(advance_state_kernel, dim3(1), dim3(64), stream_, d_state_, n_ctx_);
    return true;
}
