def patch(s):
    a = s.index("        if (wave == 3) {\n            // geometric part of dL/dbeta: sum Jd.dJ + Jdrel.drel + sel_sd.dvp with dL/drel_i = GR_p^T t_i formed inline;")
    b = s.index("            BF_MARK(52, 192, it, t_iter);\n            float *strip = S.vpp + 4 * 64;")
    new = '''        if (wave >= 1 && wave <= 3) {
            // geometric part of dL/dbeta: sum Jd.dJ + Jdrel.drel + sel_sd.dvp with dL/drel_i = GR_p^T t_i formed inline;
            // lane = (beta l = lane / 6, slice sl = lane % 6): joints sl, sl + 6, sl + 12, sl + 18 and outputs sl + 6 m.
            // Round 5: the sum is DEALT over the three waves that have nothing else in this phase's first thousand cycles - wave 1 the
            // joints 0-11 (two of the four joint rounds), wave 2 the joints 12-23, wave 3 the selector outputs - each leaves its lanes'
            // partial sums in a strip of its own, waves 1 and 2 raise a counter, wave 3 adds the three strips in wave order and steps.
            const int l = min(lane / 6, nb - 1), sl = lane - (lane / 6) * 6;
            const bool on = lane < 6 * nb;
            float acc = 0.f;
            if (wave < 3) {
                const int h = wave - 1;                  // two joints at a time (registers)
                int pj[2];
                float tv[2][3], dj[2][3], jd[2][3], jr[2][3];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int i = sl + 6 * (2 * h + m);
                    pj[m] = S.par[i];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        tv[m][k] = S.tt[i * 3 + k]; dj[m][k] = S.dJ[i * 3 + k];
                        jd[m][k] = S.Jd[(i * 3 + k) * nbp + l]; jr[m][k] = S.Jdrel[(i * 3 + k) * nbp + l];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                float4 ga[2][3];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    ga[m][0] = *(const float4 *)(S.G + pj[m] * 12); ga[m][1] = *(const float4 *)(S.G + pj[m] * 12 + 4);
                    ga[m][2] = *(const float4 *)(S.G + pj[m] * 12 + 8);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int i = sl + 6 * (2 * h + m);
                    float e0 = tv[m][0], e1 = tv[m][1], e2 = tv[m][2];
                    if (i > 0) {
                        e0 = ga[m][0].x * tv[m][0] + ga[m][1].x * tv[m][1] + ga[m][2].x * tv[m][2];
                        e1 = ga[m][0].y * tv[m][0] + ga[m][1].y * tv[m][1] + ga[m][2].y * tv[m][2];
                        e2 = ga[m][0].z * tv[m][0] + ga[m][1].z * tv[m][1] + ga[m][2].z * tv[m][2];
                    }
                    acc += jd[m][0] * dj[m][0] + jr[m][0] * e0;
                    acc += jd[m][1] * dj[m][1] + jr[m][1] * e1;
                    acc += jd[m][2] * dj[m][2] + jr[m][2] * e2;
                }
                float *mine = S.vpp + (3 + wave) * 64;     // (the strips of GMM waves 4 and 5, idle in this phase)
                mine[lane] = on ? acc : 0.f;
                BF_WAVE_FENCE();
                if (lane == 0) ((volatile int *)S.stamp)[wave] = it + 1;          // (flags in the stamp array's first slots, unused by the product build)
            } else {
            {
                float sw[6], sd[6];
#pragma unroll
                for (int m = 0; m < 6; ++m) { const int o = min(sl + 6 * m, ns3 - 1); sw[m] = S.sel_sd[o * nbp + l]; sd[m] = S.dvp[o]; }
#pragma unroll
                for (int m = 0; m < 6; ++m) acc += (sl + 6 * m < ns3) ? sw[m] * sd[m] : 0.f;
            }
'''
    s = s[:a] + new + s[b:]
    old = '''            float *strip = S.vpp + 4 * 64;                 // (the pose-blend strips of waves 0-3 are dead by now; this is a fifth)
            strip[lane] = on ? acc : 0.f;
            // this lane's beta (lanes 0..nb-1): value, moments
            const int pidx = T.off_beta + min(lane, nb - 1);
            const float pval = Pcur[pidx], am = S.am[pidx], av = S.av[pidx];
            BF_WAVE_FENCE();
            if (lane < nb) {
                float pr[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) pr[i] = strip[lane * 6 + i];
                float g = 0.f;
#pragma unroll
                for (int i = 0; i < 6; ++i) g += pr[i];'''
    new = '''            float *strip = S.vpp + 6 * 64;                 // (GMM wave 6's strip: wave 3's own partial sums)
            strip[lane] = on ? acc : 0.f;
            // this lane's beta (lanes 0..nb-1): value, moments
            const int pidx = T.off_beta + min(lane, nb - 1);
            const float pval = Pcur[pidx], am = S.am[pidx], av = S.av[pidx];
            while (((volatile int *)S.stamp)[1] < it + 1 || ((volatile int *)S.stamp)[2] < it + 1) __builtin_amdgcn_s_sleep(1);        // waves 1 and 2 have left their strips
            BF_WAVE_FENCE();
            if (lane < nb) {
                float pr[18];
#pragma unroll
                for (int w_ = 0; w_ < 3; ++w_)
#pragma unroll
                    for (int i = 0; i < 6; ++i) pr[w_ * 6 + i] = S.vpp[(4 + w_) * 64 + lane * 6 + i];
                float g = 0.f;
#pragma unroll
                for (int i = 0; i < 18; ++i) g += pr[i];'''
    assert old in s
    s = s.replace(old, new)
    # close the extra brace of the `else` branch at the end of wave 3's block
    old = '''            beta_dependent(mode == 0 ? Pnext : Pcur);
            BF_MARK(54, 192, it, t_iter);
        }
        } else {
        {
        // ================= phase I: per joint (wave 0, lane = joint)'''
    new = '''            beta_dependent(mode == 0 ? Pnext : Pcur);
            BF_MARK(54, 192, it, t_iter);
            }
        }
        } else {
        {
        // ================= phase I: per joint (wave 0, lane = joint)'''
    assert old in s
    s = s.replace(old, new)
    # transl / scale off wave 1 (it carries joints now): wave 0, lanes 32-35
    old = '''        if (tq >= 64 && tq < 68) {                                   // transl / scale
            const int pidx = tq - 64;'''
    new = '''        if (tq >= 32 && tq < 36) {                                   // transl / scale (wave 0, behind its pose lanes: waves 1-3 carry the betas)
            const int pidx = tq - 32;'''
    assert old in s
    s = s.replace(old, new)
    old = "    if (EXT && tid == 0) ((int *)S.part)[BF_POSE_STATE_FLAG] = 0;          // (the chain waves' cue: no iteration's token yet)"
    assert old in s
    s = s.replace(old, old + "\n    if (tid < 4) ((volatile int *)S.stamp)[tid] = 0;                          // (the betas' join flags of the Adam phase)")
    return s
