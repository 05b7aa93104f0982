// legacy GGML graphs (gpt2 / starcoder: gpt2.cc:391-699; mpt: mpt.cc:365-590): prompt chunks and the token steps — part of engine.cc (one translation unit: the HIP kernels are templates and file-local helpers of it); included there,
// inside namespace ctamd, after the launch helpers it uses.  Not a stand-alone header.

// gpt2_eval (models/llms/gpt2.cc:391-699) for the nt tokens of a chunk: token_step_gpt2's launches over rows of the chunk; the
// K / V rows of all its tokens are appended to the F32 cache before the attention launch.
bool Engine::chunk_step_gpt2(int nt, bool want_logits, std::string& err) {
    const int E = hp_.n_embd, F = hp_.n_ff, hd = hp_.head_dim();
    const int* d_pos = d_state_ + 1;
    CT_LAUNCH(embed_row_kernel, dim3((unsigned)std::max(1, E / 256), (unsigned)nt), dim3(256), stream_, tok_embd_.raw, tok_embd_.type, E,
              (const int*)d_tokens_, (const int*)d_state_, xb_, (const float*)wpe_);
    MatvecArgs base = MatvecArgs();
    base.pos = d_pos;
    base.n_ctx = n_ctx_;
    base.head_dim = hd;
    base.n_embd_gqa = E;
    base.v_stride = v_stride_;
    base.silu_tab = silu_tab_;
    base.gelu_tab = gelu_tab_;
    base.eps = hp_.rms_eps;
    const float kq_scale = (float)(1.0 / sqrt((double)((float)E / (float)hp_.n_head)));   // gpt2.cc:540-543 (see launch_attention)
    for (int il = 0; il < hp_.n_layer; ++il) {
        const Layer& L = layers_[il];
        float* km = kmem_ + (size_t)il * n_ctx_ * E;
        float* vm = vmem_ + (size_t)il * n_ctx_ * E;
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.norm_w = L.attn_norm; a.norm_b = L.attn_norm_b;
            a.out = qkv_tmp_b_; a.bias = L.b_qkv;
            set_jobs(a, {{&L.wqkv, EPI_BIAS_STORE}});
            if (!pf_matvec(a, xb_, E, nt, 3 * E, 0, "qkv", (double)L.wqkv.bytes, err)) return false;
        }
        CT_LAUNCH(gpt2_kv_append_kernel, dim3((unsigned)nt), dim3(256), stream_, (const float*)qkv_tmp_b_, km, vm, d_pos, E);
        CT_LAUNCH(attn_f32_exact_kernel, dim3((unsigned)hp_.n_head, (unsigned)nt), dim3(256), stream_, (const float*)qkv_tmp_b_, km, vm,
                  attn_out_b_, (const uint16_t*)exp_tab_, d_pos, (const int*)(d_state_ + 2), E, hd, kq_scale, 0);
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_PLAIN; a.out = xb_; a.res = xb_; a.bias = L.b_wo;
            set_jobs(a, {{&L.wo, EPI_BIAS_ADD}});
            if (!pf_matvec(a, attn_out_b_, E, nt, E, E, "wo", (double)L.wo.bytes, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.norm_w = L.ffn_norm; a.norm_b = L.ffn_norm_b; a.out = hb_; a.bias = L.b_up;
            set_jobs(a, {{&L.w_up, EPI_BIAS_GELU}});
            if (!pf_matvec(a, xb_, E, nt, F, 0, "ffn_up", (double)L.w_up.bytes, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = F; a.pro = PRO_PLAIN; a.out = xb_; a.res = xb_; a.bias = L.b_down;
            set_jobs(a, {{&L.w_down, EPI_BIAS_ADD}});
            if (!pf_matvec(a, hb_, F, nt, E, E, "down", (double)L.w_down.bytes, err)) return false;
        }
    }
    if (want_logits) {
        MatvecArgs a = base;
        a.K = E; a.pro = PRO_LAYERNORM; a.x = xb_ + (size_t)(nt - 1) * E; a.norm_w = output_norm_; a.norm_b = output_norm_b_; a.out = d_logits_;
        set_jobs(a, {{&output_, EPI_STORE}});
        if (!run_matvec(a, err)) return false;
    }
    CT_LAUNCH(advance_state_n_kernel, dim3(1), dim3(64), stream_, d_state_, nt);
    return true;
}


// mpt_eval for the nt tokens of a chunk: token_step_mpt's launches over the rows of the chunk.
bool Engine::chunk_step_mpt(int nt, bool want_logits, std::string& err) {
    const int E = hp_.n_embd, F = hp_.n_ff, hd = hp_.head_dim();
    const int* d_pos = d_state_ + 1;
    CT_LAUNCH(embed_row_kernel, dim3((unsigned)std::max(1, E / 256), (unsigned)nt), dim3(256), stream_, tok_embd_.raw, tok_embd_.type, E,
              (const int*)d_tokens_, (const int*)d_state_, xb_);
    MatvecArgs base = MatvecArgs();
    base.pos = d_pos;
    base.n_ctx = n_ctx_;
    base.head_dim = hd;
    base.n_embd_gqa = E;
    base.v_stride = v_stride_;
    base.silu_tab = silu_tab_;
    base.gelu_tab = gelu_tab_;
    base.eps = hp_.rms_eps;
    for (int il = 0; il < hp_.n_layer; ++il) {
        const Layer& L = layers_[il];
        uint16_t* kc = kcache_ + (size_t)il * n_ctx_ * E;
        uint16_t* vc = vcache_ + (size_t)il * v_stride_ * E;
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.norm_w = L.attn_norm; a.norm_b = L.attn_norm_b; a.out = qkv_tmp_b_;
            set_jobs(a, {{&L.wqkv, EPI_STORE}});
            if (!pf_matvec(a, xb_, E, nt, 3 * E, 0, "qkv", (double)L.wqkv.bytes, err)) return false;
        }
        CT_LAUNCH(mpt_store_kernel, dim3((unsigned)(3 * hp_.n_head), (unsigned)nt), dim3((unsigned)hd), stream_, (const float*)qkv_tmp_b_, q_f16_b_,
                  kc, vc, d_pos, hp_.n_head, hd, n_ctx_, v_stride_, clip_qkv_);
        launch_attention(kc, vc, nt);
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_PLAIN; a.out = xb_; a.res = xb_;
            set_jobs(a, {{&L.wo, EPI_ADD}});
            if (!pf_matvec(a, attn_out_b_, E, nt, E, E, "wo", (double)L.wo.bytes, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.norm_w = L.ffn_norm; a.norm_b = L.ffn_norm_b; a.out = hb_;
            set_jobs(a, {{&L.w_up, EPI_GELU}});
            if (!pf_matvec(a, xb_, E, nt, F, 0, "ffn_up", (double)L.w_up.bytes, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = F; a.pro = PRO_PLAIN; a.out = xb_; a.res = xb_;
            set_jobs(a, {{&L.w_down, EPI_ADD}});
            if (!pf_matvec(a, hb_, F, nt, E, E, "down", (double)L.w_down.bytes, err)) return false;
        }
    }
    if (want_logits) {
        MatvecArgs a = base;
        a.K = E; a.pro = PRO_LAYERNORM; a.x = xb_ + (size_t)(nt - 1) * E; a.norm_w = output_norm_; a.norm_b = output_norm_b_; a.out = d_logits_;
        set_jobs(a, {{&output_, EPI_STORE}});
        if (!run_matvec(a, err)) return false;
    }
    CT_LAUNCH(advance_state_n_kernel, dim3(1), dim3(64), stream_, d_state_, nt);
    return true;
}

// gpt2_eval (models/llms/gpt2.cc:391-699), one token: wte + wpe, then per layer
//   LN(ln_1) -> Q8_0 -> c_attn + b -> [append K,V rows to the F32 cache] -> F32 attention -> c_proj + b -> + x
//   LN(ln_2) -> Q8_0 -> c_fc + b -> GELU table -> Q8_0 -> c_proj + b -> + x;   final LN -> lm_head (tied wte)
bool Engine::token_step_gpt2(bool want_logits, std::string& err) {
    const int E = hp_.n_embd, F = hp_.n_ff, hd = hp_.head_dim();
    const int* d_pos = d_state_ + 1;
    CT_LAUNCH(embed_row_kernel, dim3((unsigned)std::max(1, E / 256)), dim3(256), stream_, tok_embd_.raw, tok_embd_.type, E,
              (const int*)d_tokens_, (const int*)d_state_, x_, (const float*)wpe_);
    MatvecArgs base = MatvecArgs();
    base.pos = d_pos;
    base.n_ctx = n_ctx_;
    base.head_dim = hd;
    base.n_embd_gqa = E;
    base.v_stride = v_stride_;
    base.silu_tab = silu_tab_;
    base.gelu_tab = gelu_tab_;
    base.eps = hp_.rms_eps;
    base.dbg_sink = scores_; base.f16_tmp = f16_tmp_;
    const float kq_scale = (float)(1.0 / sqrt((double)((float)E / (float)hp_.n_head)));   // gpt2.cc:540-543 (see launch_attention)
    for (int il = 0; il < hp_.n_layer; ++il) {
        const Layer& L = layers_[il];
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.x = x_; a.norm_w = L.attn_norm; a.norm_b = L.attn_norm_b;
            a.out = qkv_tmp_; a.bias = L.b_qkv;
            set_jobs(a, {{&L.wqkv, EPI_BIAS_STORE}});
            if (!run_matvec(a, err)) return false;
        }
        CT_LAUNCH(attn_f32_exact_kernel, dim3((unsigned)hp_.n_head), dim3(256), stream_, (const float*)qkv_tmp_,
                  kmem_ + (size_t)il * n_ctx_ * E, vmem_ + (size_t)il * n_ctx_ * E, attn_out_, (const uint16_t*)exp_tab_, d_pos,
                  (const int*)(d_state_ + 2), E, hd, kq_scale);
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_PLAIN; a.x = attn_out_; a.out = x_; a.res = x_; a.bias = L.b_wo;
            set_jobs(a, {{&L.wo, EPI_BIAS_ADD}});
            if (!run_matvec(a, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.x = x_; a.norm_w = L.ffn_norm; a.norm_b = L.ffn_norm_b; a.out = h_; a.bias = L.b_up;
            set_jobs(a, {{&L.w_up, EPI_BIAS_GELU}});
            if (!run_matvec(a, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = F; a.pro = PRO_PLAIN; a.x = h_; a.out = x_; a.res = x_; a.bias = L.b_down;
            set_jobs(a, {{&L.w_down, EPI_BIAS_ADD}});
            if (!run_matvec(a, err)) return false;
        }
    }
    if (want_logits) {
        MatvecArgs a = base;
        a.K = E; a.pro = PRO_LAYERNORM; a.x = x_; a.norm_w = output_norm_; a.norm_b = output_norm_b_; a.out = d_logits_;
        set_jobs(a, {{&output_, EPI_STORE}});
        if (!run_matvec(a, err)) return false;
    }
    CT_LAUNCH(advance_state_kernel, dim3(1), dim3(64), stream_, d_state_, n_ctx_);
    return true;
}


// mpt_eval (models/llms/mpt.cc:365-590), one token: wte row, then per layer
//   norm * ln_1 -> Q8_0 -> Wqkv -> clamp -> fp16 Q / K / V -> fp16 attention with the ALiBi term -> out_proj -> + x
//   norm * ln_2 -> Q8_0 -> up_proj -> GELU table -> Q8_0 -> down_proj -> + x;   final norm * norm_f -> wte as the head
bool Engine::token_step_mpt(bool want_logits, std::string& err) {
    const int E = hp_.n_embd, F = hp_.n_ff, hd = hp_.head_dim();
    const int* d_pos = d_state_ + 1;
    CT_LAUNCH(embed_row_kernel, dim3((unsigned)std::max(1, E / 256)), dim3(256), stream_, tok_embd_.raw, tok_embd_.type, E,
              (const int*)d_tokens_, (const int*)d_state_, x_);
    MatvecArgs base = MatvecArgs();
    base.pos = d_pos;
    base.n_ctx = n_ctx_;
    base.head_dim = hd;
    base.n_embd_gqa = E;
    base.v_stride = v_stride_;
    base.silu_tab = silu_tab_;
    base.gelu_tab = gelu_tab_;
    base.eps = hp_.rms_eps;
    base.dbg_sink = scores_; base.f16_tmp = f16_tmp_;
    for (int il = 0; il < hp_.n_layer; ++il) {
        const Layer& L = layers_[il];
        uint16_t* kc = kcache_ + (size_t)il * n_ctx_ * E;
        uint16_t* vc = vcache_ + (size_t)il * v_stride_ * E;
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.x = x_; a.norm_w = L.attn_norm; a.norm_b = L.attn_norm_b; a.out = qkv_tmp_;
            set_jobs(a, {{&L.wqkv, EPI_STORE}});
            if (!run_matvec(a, err)) return false;
        }
        CT_LAUNCH(mpt_store_kernel, dim3((unsigned)(3 * hp_.n_head)), dim3((unsigned)hd), stream_, (const float*)qkv_tmp_, q_f16_, kc, vc, d_pos,
                  hp_.n_head, hd, n_ctx_, v_stride_, clip_qkv_);
        launch_attention(kc, vc);
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_PLAIN; a.x = attn_out_; a.out = x_; a.res = x_;
            set_jobs(a, {{&L.wo, EPI_ADD}});
            if (!run_matvec(a, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = E; a.pro = PRO_LAYERNORM; a.x = x_; a.norm_w = L.ffn_norm; a.norm_b = L.ffn_norm_b; a.out = h_;
            set_jobs(a, {{&L.w_up, EPI_GELU}});
            if (!run_matvec(a, err)) return false;
        }
        {
            MatvecArgs a = base;
            a.K = F; a.pro = PRO_PLAIN; a.x = h_; a.out = x_; a.res = x_;
            set_jobs(a, {{&L.w_down, EPI_ADD}});
            if (!run_matvec(a, err)) return false;
        }
    }
    if (want_logits) {
        MatvecArgs a = base;
        a.K = E; a.pro = PRO_LAYERNORM; a.x = x_; a.norm_w = output_norm_; a.norm_b = output_norm_b_; a.out = d_logits_;
        set_jobs(a, {{&output_, EPI_STORE}});
        if (!run_matvec(a, err)) return false;
    }
    CT_LAUNCH(advance_state_kernel, dim3(1), dim3(64), stream_, d_state_, n_ctx_);
    return true;
}
