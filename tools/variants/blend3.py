def patch(s):
    # 1. GMM waves 4-6 no longer blend (wave 3 does their nine vertices, below); wave 7 keeps vertices 9, 10
    old = "            if (merge_bc) { gmm_blend(); BF_MARK(42, 256, it, t_iter); }"
    new = "            if (merge_bc && gwi == 3) { gmm_blend(); BF_MARK(42, 256, it, t_iter); }      // (vertices 9.. : the others are wave 3's, wave3_blend)"
    assert old in s
    s = s.replace(old, new)
    # 2. wave 3's blend, defined next to gmm_blend
    old = "    // d(pose feature) = sel_pd . dvp in phase F, a quarter of the rows per GMM wave"
    new = '''    // Round 5: the pose blend of selector vertices 0-8 on WAVE 3, which has nothing else in phase A (its SIMD carries ~1,600 instructions per
    // iteration against ~2,100 on the chain waves' SIMDs, where GMM waves 4-6 did these blends): lane = (output o = lane / 2 of 27, half h
    // of the 210 table rows: 104 each, two rows past the pose feature), b64 pairs from the transposed table, the two halves meet in one DPP add.
    auto wave3_blend = [&]() {
        typedef float f2 __attribute__((ext_vector_type(2)));
        constexpr int NP = 52, NB4 = 4, PB = NP / NB4;         // 52 row pairs per lane in four batches of thirteen
        constexpr int NPF = NJ > 0 ? 9 * (NJ - 1) : 8;
        const int lq = bf_launder(lane);
        const int o = min(lq >> 1, 26), h = lq & 1;
        const bool on = (lq >> 1) < 27 && o < ns3;
        const int p0 = h * 104;
        const f2 *fp = (const f2 *)__builtin_assume_aligned(S.feat + p0, 8);
        const f2 *wp_ = (const f2 *)__builtin_assume_aligned(S.sel_pd2 + o * BF_PDT_LD + p0, 8);
        float acc = 0.f;
#pragma unroll
        for (int b = 0; b < NB4; ++b) {
            f2 f[PB], w[PB];
#pragma unroll
            for (int i = 0; i < PB; ++i) { f[i] = fp[b * PB + i]; w[i] = wp_[b * PB + i]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int pl = 2 * (b * PB + i);              // (row inside the half: only the second half's last pair reaches past the feature)
                acc += f[i].x * w[i].x;
                acc += ((pl + 1 + 104 < NPF) || h == 0 ? f[i].y : 0.f) * w[i].y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        acc = dpp_add<0xB1>(acc);                              // the other half (lane ^ 1)
        // (rows 208, 209 of the table are zero padding: nobody reads them)
        if (on && h == 0) S.vp[o] = S.vs[o] + acc;
    };
    // d(pose feature) = sel_pd . dvp in phase F, a quarter of the rows per GMM wave'''
    assert old in s
    s = s.replace(old, new, 1)
    # 3. wave 3 calls it in phase A (before the door work of the dense schedule)
    old = '''        } else if (EXT && door) {         // wave 3 has nothing of its own in this phase
            door_token = it + 1;
            door_state(Pcur);
            BF_MARK(59, 192, it, t_iter);
        }'''
    new = '''        } else {                          // wave 3
            if (merge_bc) wave3_blend();
            if (EXT && door) {
                door_token = it + 1;
                door_state(Pcur);
                BF_MARK(59, 192, it, t_iter);
            }
        }'''
    assert old in s
    s = s.replace(old, new)
    return s
