def patch(s):
    a = s.index("        for (int it = 0; it < n_iters; ++it) {\n#ifdef BF_STAMP\n            int sidx = 0;\n            const long long t_iter = 0;\n#else\n            const long long t_iter = 0;\n#endif\n            // The GMM prior (d = theta - mu")
    b = s.index("            constexpr bool GMM_FG = NJ > 0 && NJ <= 32 && NS > 0 && NS * 3 <= 36;     // (= MERGE_FG of the geometry loop)")
    new = '''        constexpr bool GMM_AHEAD = GBLEND;
        constexpr int NAH = 6;
        v2f dA = {0.f, 0.f}, dB = {0.f, 0.f};
        v2f y0 = {0.f, 0.f}, y1 = {0.f, 0.f};
        auto gmm_load_d = [&](const float *P) {
            const int lq0 = bf_launder(lane);
            const int src0 = lq0 < T.nbp ? T.off_pose + lq0 : -1, src1 = 64 + lq0 < T.nbp ? T.off_pose + 64 + lq0 : -1;
            if (src0 >= 0) { const float th = P[src0]; dA.x = th - gd_mu[0]; dA.y = th - gd_mu[1]; }
            else { dA.x = -gd_mu[0]; dA.y = -gd_mu[1]; }
            dB.x = 0.f; dB.y = 0.f;
            if (lane < BF_GMM_D - 64) {
                const float th = src1 >= 0 ? P[src1] : 0.f;
                dB.x = th - gd_mu[2]; dB.y = th - gd_mu[3];
            }
            ((v2f *)gdw)[lane] = dA;
            if (lane < BF_GMM_LD - 64) ((v2f *)gdw)[64 + lane] = dB;
            BF_WAVE_FENCE();
        };
#define BF_GMM_CHUNK(c)                                                                         \\
            _Pragma("unroll") for (int j2 = 4 * (c); j2 < 4 * (c) + 4; ++j2) {                  \\
                const float4 t = d4[j2];                                                        \\
                const v2f t0 = {t.x, t.y}, t1 = {t.z, t.w};                                     \\
                if (2 * j2 < (GBLEND ? BF_GMM_D : BF_GMM_LD)) y0 += P2[2 * j2] * t0;            \\
                if (2 * j2 + 1 < (GBLEND ? BF_GMM_D : BF_GMM_LD)) y1 += P2[2 * j2 + 1] * t1;    \\
            }
#define BF_GMM_PIPELINE(N)                                                                      \\
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                  \\
            _Pragma("unroll") for (int c_ = 0; c_ < (N) - 1; ++c_) {                            \\
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                              \\
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                              \\
            }                                                                                   \\
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        auto gmm_ahead = [&](const float *P) {
            gmm_load_d(P);
            y0 = v2f{0.f, 0.f}; y1 = v2f{0.f, 0.f};
            BF_GMM_CHUNK(0)
            BF_GMM_CHUNK(1)
            BF_GMM_CHUNK(2)
            BF_GMM_CHUNK(3)
            BF_GMM_CHUNK(4)
            BF_GMM_CHUNK(5)
            BF_GMM_PIPELINE(6)
        };
        static_assert(NAH == 6, "gmm_ahead spells its chunks out");
        if (GMM_AHEAD) gmm_ahead(Pcur);
        for (int it = 0; it < n_iters; ++it) {
#ifdef BF_STAMP
            int sidx = 0;
            const long long t_iter = 0;
#else
            const long long t_iter = 0;
#endif
            if (!GMM_AHEAD) { gmm_load_d(Pcur); y0 = v2f{0.f, 0.f}; y1 = v2f{0.f, 0.f}; }
            if (merge_bc) { gmm_blend(); BF_MARK(42, 256, it, t_iter); }
'''
    s = s[:a] + new + s[b:]
    a = s.index("            BF_GMM_CHUNK(0)\n            if (!GBLEND) {\n                BF_GMM_CHUNK(1)\n                BF_GMM_CHUNK(2)\n            }\n            BF_SYNC();                 // A")
    b = s.index("            if (!NO_VERT && !MERGE_BD) BF_SYNC();   // C")
    new = '''            if (GMM_AHEAD) {
                BF_GMM_CHUNK(6)
                BF_GMM_CHUNK(7)
                BF_GMM_CHUNK(8)
                BF_GMM_PIPELINE(3)
            } else {
                BF_GMM_CHUNK(0)
                if (!GBLEND) {
                    BF_GMM_CHUNK(1)
                    BF_GMM_CHUNK(2)
                }
            }
            BF_SYNC();                 // A
            if (EXT && door) door_mid(it, Pcur, std::false_type());
            if (!merge_bc) {           // (two-phase pose blend: everybody takes row slices)
                pose_blend(std::integral_constant<int, 2>());
                if (!NO_VERT) BF_SYNC();             // B
            }
            if (GMM_AHEAD) { }
            else if (GBLEND) {
                BF_GMM_CHUNK(1)
                BF_GMM_CHUNK(2)
                BF_GMM_CHUNK(3)
                BF_GMM_CHUNK(4)
                BF_GMM_CHUNK(5)
                BF_GMM_CHUNK(6)
                BF_GMM_CHUNK(7)
                BF_GMM_CHUNK(8)
                BF_GMM_PIPELINE(8)
                BF_MARK(49, 256, it, t_iter);
            } else {
                BF_GMM_CHUNK(3)
                BF_GMM_CHUNK(4)
                BF_GMM_CHUNK(5)
            }
'''
    s = s[:a] + new + s[b:]
    old = '''            BF_SYNC();                 // K
            if (mode == 0) { float *sw = Pcur; Pcur = Pnext; Pnext = sw; }
        }
    } else {'''
    new = '''            if (GMM_AHEAD && mode == 0 && it + 1 < n_iters) {
                while (*(volatile int *)S.scal < it + 1) __builtin_amdgcn_s_sleep(1);
                BF_WAVE_FENCE();
                gmm_ahead(Pnext);
            }
            BF_SYNC();                 // K
            if (mode == 0) { float *sw = Pcur; Pcur = Pnext; Pnext = sw; }
        }
#undef BF_GMM_PIPELINE
    } else {'''
    assert old in s
    s = s.replace(old, new)
    old = '''                    for (int c = 0; c < 3; ++c) { pn[c] = pv[c] - at1 * (pm[c] * rd[c]); Pnext[pi0 + c] = pn[c]; S.am[pi0 + c] = pm[c]; S.av[pi0 + c] = pw[c]; }'''
    new = old + '''
                    if (GBLEND) { BF_WAVE_FENCE(); if (tq == 0) *(volatile int *)S.scal = it + 1; }'''
    assert old in s
    s = s.replace(old, new)
    old = "    if (EXT && tid == 0) ((int *)S.part)[BF_POSE_STATE_FLAG] = 0;          // (the chain waves' cue: no iteration's token yet)"
    assert old in s
    s = s.replace(old, old + "\n    if (tid == 0) *(volatile int *)S.scal = 0;")
    return s
