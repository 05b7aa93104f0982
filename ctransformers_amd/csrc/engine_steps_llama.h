// llama graph (llm_build_llama, llama.cpp:2162-2491): prompt chunks and the token step — part of engine.cc (one translation unit: the HIP kernels are templates and file-local helpers of it); included there,
// inside namespace ctamd, after the launch helpers it uses.  Not a stand-alone header.

// llm_build_llama (llama.cpp:2162-2491) for nt tokens of one batch_eval chunk at once: the same launches as token_step,
// each over rows [c0, c0 + nt) of the chunk (kernels_pf.h).  The cursor in d_state_ is at token c0 on entry.
bool Engine::chunk_step(int c0, int nt, bool want_logits, std::string& err) {
    if (hp_.falcon()) return chunk_step_falcon(c0, nt, want_logits, err);
    if (hp_.gpt2()) return chunk_step_gpt2(nt, want_logits, err);
    if (hp_.mpt()) return chunk_step_mpt(nt, want_logits, err);
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
    base.eps = hp_.rms_eps;
    for (int il = l0_; il < l1_; ++il) {
        const Layer& L = layers_[il];
        cur_layer_ = il;
        uint16_t* kc = kcache_ + (size_t)(il - l0_) * n_ctx_ * G;
        uint16_t* vc = vcache_ + (size_t)(il - l0_) * v_stride_ * G;
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_RMSNORM; a.norm_w = L.attn_norm;
            a.q_f16 = q_f16_b_; a.kcache = kc; a.vcache = vc;
            set_jobs(a, {{&L.wq, EPI_ROPE_Q}, {&L.wk, EPI_ROPE_K}, {&L.wv, EPI_V}});
            if (!pf_matvec(a, xb_, E, nt, 0, 0, "qkv", (double)(L.wq.bytes + L.wk.bytes + L.wv.bytes), err)) return false;
        }
        if (site_on("attn_fused")) {
            prof_begin("attn_fused", "attn_fused_exact_kernel", 0.0);
            launch_attention(kc, vc, nt);
            prof_end();
        }
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_PLAIN; a.out = xb_; a.res = xb_;
            set_jobs(a, {{&L.wo, EPI_ADD}});
            if (!pf_matvec(a, attn_out_b_, E, nt, E, E, "wo", (double)L.wo.bytes, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_RMSNORM; a.norm_w = L.ffn_norm; a.out = hb_;
            a.job[0].w = L.w_gate; a.job[0].pair0 = 0; a.job[0].epi = EPI_SILU_MUL;
            a.job[1].w = L.w_up; a.job[1].pair0 = 0; a.job[1].epi = EPI_SILU_MUL;
            a.njobs = 2; a.gateup = 1; a.n_pairs = F;
            if (L.w_gu.r2) { a.job[0].w = L.w_gu; a.njobs = 1; }   // K-quants: the fused matrix of the decode path (kernels_pg.h)
            else if (fast_pf_ && L.w_gu.m8) { a.job[0].w = L.w_gu; a.njobs = 1; }   // Q8_0: the fused LAYOUT_M8 matrix (kernels_mm8.h)
            if (!pf_matvec(a, xb_, E, nt, F, 0, "gate_up", (double)(L.w_gate.bytes + L.w_up.bytes), err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = F; a.pro = PRO_PLAIN; a.out = xb_; a.res = xb_;
            set_jobs(a, {{&L.w_down, EPI_ADD}});
            if (!pf_matvec(a, hb_, F, nt, E, E, "down", (double)L.w_down.bytes, err)) return false;
        }
    }
    if (l1_ < hp_.n_layer) {
        HIP_OK(hipMemcpyAsync(xio_ + (size_t)c0 * E, xb_, (size_t)nt * E * 4, hipMemcpyDeviceToDevice, stream_));
    } else if (want_logits) {   // the chunk's last token only (llama.cpp:2955-2959 keeps the last column)
        const float* xl = xb_ + (size_t)(nt - 1) * E;
        MatvecArgs a = base;
        a.K = E; a.pro = PRO_RMSNORM; a.x = xl; a.norm_w = output_norm_; a.out = d_logits_;
        set_jobs(a, {{&output_, EPI_STORE}});
        if (kq_can(a)) { a.emb_out = d_emb_; set_head_fold(a, false); }   // the head launch stores the final-norm output from its prologue and picks the greedy token
        else CT_LAUNCH((rmsnorm_f32_kernel<256>), dim3(1), dim3(256), stream_, xl, (const float*)output_norm_, d_emb_, E, hp_.rms_eps);
        if (!run_matvec(a, err)) return false;
        if (a.pick_ws) launch_pick(a);
    }
    CT_LAUNCH(advance_state_n_kernel, dim3(1), dim3(64), stream_, d_state_, nt);
    return true;
}


bool Engine::token_step(bool want_logits, std::string& err) {
    if (hp_.falcon()) return token_step_falcon(want_logits, err);
    if (hp_.gpt2()) return token_step_gpt2(want_logits, err);
    if (hp_.mpt()) return token_step_mpt(want_logits, err);
    const int E = hp_.n_embd, G = hp_.n_embd_gqa(), F = hp_.n_ff, hd = hp_.head_dim();
    const int* d_pos = d_state_ + 1;
    if (stamps_) CT_LAUNCH(stamp_kernel, dim3(1), dim3(1), stream_, stamps_, 1ull);
    if (l0_ == 0 && cont_mode_) {
        // continuation step of a greedy chain: the previous head launch left the embedding row of its pick in x_
    } else if (l0_ == 0) {
        if (site_on("embed")) {
        prof_begin("embed", "embed_row_kernel", (double)ggml_row_bytes(tok_embd_.type, E));
        CT_LAUNCH(embed_row_kernel, dim3((unsigned)std::max(1, E / 256)), dim3(256), stream_, tok_embd_.raw, tok_embd_.type, E,
                  (const int*)d_tokens_, (const int*)d_state_, x_);
        prof_end();
        }
    } else {  // inner stage: this token's residual-stream row was handed over by the previous stage
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
    base.eps = hp_.rms_eps;
    base.dbg = env_int("CT_AMD_DBG", 0);
    base.dbg_sink = scores_; base.f16_tmp = f16_tmp_;
    bool bumped = false;
    for (int il = l0_; il < l1_; ++il) {
        const Layer& L = layers_[il];
        cur_layer_ = il;
        uint16_t* kc = kcache_ + (size_t)(il - l0_) * n_ctx_ * G;
        uint16_t* vc = vcache_ + (size_t)(il - l0_) * v_stride_ * G;
        {   // RMSNorm -> Q8_K -> {Wq,Wk,Wv} -> RoPE -> fp16 Q / KV-cache append [-> attention, where the fused launch applies]
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_RMSNORM; a.x = x_; a.norm_w = L.attn_norm;
            a.q_f16 = q_f16_; a.kcache = kc; a.vcache = vc;
            set_jobs(a, {{&L.wq, EPI_ROPE_Q}, {&L.wk, EPI_ROPE_K}, {&L.wv, EPI_V}});  // types may differ per matrix
            apply_trace(a, "qkv");
            const bool fused = qa_can(L) && !only_site_ && !prof_ && (!trace_site_ || !strcmp(trace_site_, "qa"));
            if (fused) {
                if (trace_site_) { a.dbg |= 32; a.dbg_sink = (float*)(trace_buf_ + 256); }   // (ctamd_trace_site("qa"): the mat-vec phase's stamps behind the attention phase's)
                if (!launch_qkv_attn(a, kc, vc, il, err)) return false;
                debug_dump("2attn", il);
                if (stamps_ && stamps_level_ > 1) CT_LAUNCH(stamp_kernel, dim3(1), dim3(1), stream_, stamps_, 4ull);
            } else {
                if (site_on("qkv")) {
                    prof_begin("qkv", "matvec", (double)(L.wq.bytes + L.wk.bytes + L.wv.bytes));
                    if (!run_matvec(a, err)) return false;
                    prof_end();
                }
                debug_dump("1qkv", il);
                if (stamps_ && stamps_level_ > 1) CT_LAUNCH(stamp_kernel, dim3(1), dim3(1), stream_, stamps_, 3ull);
                if (site_on("attn_fused")) {
                    prof_begin("attn_fused", "attn_fused_exact_kernel", 0.0);
                    launch_attention(kc, vc);
                    prof_end();
                }
                debug_dump("2attn", il);
                if (stamps_ && stamps_level_ > 1) CT_LAUNCH(stamp_kernel, dim3(1), dim3(1), stream_, stamps_, 4ull);
            }
        }
        {   // Q8_K(attn) -> Wo -> + residual
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_PLAIN; a.x = attn_out_; a.out = x_; a.res = x_;
            set_jobs(a, {{&L.wo, EPI_ADD}});
            apply_trace(a, "wo");
            if (site_on("wo")) {
                prof_begin("wo", "matvec", (double)L.wo.bytes);
                if (!run_matvec(a, err)) return false;
                prof_end();
            }
            debug_dump("3wo", il);
            if (stamps_ && stamps_level_ > 1) CT_LAUNCH(stamp_kernel, dim3(1), dim3(1), stream_, stamps_, 5ull);
        }
        {   // RMSNorm -> Q8_K -> {W_gate, W_up} -> SiLU(gate)*up
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_RMSNORM; a.x = x_; a.norm_w = L.ffn_norm; a.out = h_;
            if (L.w_gu.r9) {   // one job, the fused matrix (LAYOUT_L9 arena)
                a.job[0].w = L.w_gu; a.job[0].pair0 = 0; a.job[0].epi = EPI_SILU_MUL;
                a.njobs = 1; a.gateup = 1; a.n_pairs = F;
            } else {
                a.job[0].w = L.w_gate; a.job[0].pair0 = 0; a.job[0].epi = EPI_SILU_MUL;
                a.job[1].w = L.w_up; a.job[1].pair0 = 0; a.job[1].epi = EPI_SILU_MUL;
                a.njobs = 2; a.gateup = 1; a.n_pairs = F;
            }
            apply_trace(a, "gate_up");
            if (site_on("gate_up")) {
                prof_begin("gate_up", "matvec", (double)(L.w_gate.bytes + L.w_up.bytes));
                if (!run_matvec(a, err)) return false;
                prof_end();
            }
            debug_dump("4gateup", il);
            if (stamps_ && stamps_level_ > 1) CT_LAUNCH(stamp_kernel, dim3(1), dim3(1), stream_, stamps_, 6ull);
        }
        {   // Q8_K(h) -> W_down -> + residual
            MatvecArgs a = base;
            a.K = F; a.pro = PRO_PLAIN; a.x = h_; a.out = x_; a.res = x_;
            set_jobs(a, {{&L.w_down, EPI_ADD}});
            if (il == hp_.n_layer - 1 && !only_site_ && !prof_ && kq_can(a)) { a.bump = d_state_; bumped = true; }   // the token's last launch that does not read the cursor advances it
            apply_trace(a, "down");
            if (site_on("down")) {
                prof_begin("down", "matvec_k12288", (double)L.w_down.bytes);
                if (!run_matvec(a, err)) return false;
                prof_end();
            }
            debug_dump("5down", il);
            if (stamps_ && stamps_level_ > 1) CT_LAUNCH(stamp_kernel, dim3(1), dim3(1), stream_, stamps_, 7ull);
        }
    }
    if (l1_ < hp_.n_layer) {  // hand this token's residual-stream row to the next stage
        CT_LAUNCH(stage_row_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), stream_, (const float*)x_, xio_, E,
                  (const int*)d_state_, 1);
    } else if (want_logits) {
        MatvecArgs a = base;
        a.K = E; a.pro = PRO_RMSNORM; a.x = x_; a.norm_w = output_norm_; a.out = d_logits_;
        set_jobs(a, {{&output_, EPI_STORE}});
        if (bumped) a.pos = nullptr;   // the cursor was advanced by the last layer's ffn_down launch: nothing of the head launch may read it
        if (kq_can(a)) { a.emb_out = d_emb_; set_head_fold(a, bumped); }   // the head launch stores the final-norm output from its prologue, picks the greedy token and prepares the next step
        else if (!only_site_)
            CT_LAUNCH((rmsnorm_f32_kernel<256>), dim3(1), dim3(256), stream_, (const float*)x_, (const float*)output_norm_, d_emb_, E,
                      hp_.rms_eps);
        apply_trace(a, "lm_head");
        if (site_on("lm_head")) {
            prof_begin("lm_head", "matvec", (double)output_.bytes);
            if (!run_matvec(a, err)) return false;
            prof_end();
        }
        if (stamps_ && stamps_level_ > 1) CT_LAUNCH(stamp_kernel, dim3(1), dim3(1), stream_, stamps_, 8ull);
        if (a.pick_ws && !only_site_) launch_pick(a);
    }
    if (!only_site_ && !bumped) CT_LAUNCH(advance_state_kernel, dim3(1), dim3(64), stream_, d_state_, n_ctx_);
    if (stamps_) CT_LAUNCH(stamp_kernel, dim3(1), dim3(1), stream_, stamps_, 2ull);
    if (dump_dir_) ++dump_seq_;
    return true;
}
