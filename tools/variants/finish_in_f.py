def patch(s):
    # 1. the GMM waves finish the prior in phase F and share its gradient terms among themselves (a counter in LDS joins the four waves)
    old = '''            if (GBLEND) {              // D (+E): projection, view reduction and routing on the geometry waves: the rest of the prior
                gmm_finish();
                BF_MARK(55, 256, it, t_iter);
            }
            if (!NO_VERT) BF_SYNC();   // D'''
    new = '''            // (GBLEND, round 5: the rest of the prior LEFT this phase - the GMM waves were its pole, 3,100-3,400 cycles against the geometry
            //  waves' 2,450-2,750 - for phase F, where they had 1,000 cycles to spare)
            if (!NO_VERT) BF_SYNC();   // D'''
    assert old in s
    s = s.replace(old, new)
    old = '''            } else {                   // F: this wave's quarter of d(pose feature) = sel_pd . dvp
                if (merge_bc) gmm_dfeat();
            }'''
    new = '''            } else {
                // F: the rest of the prior (y -> LDS, tail pieces, the two quadratic forms), then this wave's quarter of d(pose feature)
                // = sel_pd . dvp; then - once all four GMM waves have left their q values (a counter in S.scal[0]: the waves of the other
                // role are not involved, so no workgroup barrier) - this wave's share of the priors' gradient terms for the Adam phase
                gmm_finish();
                BF_WAVE_FENCE();
                if (lane == 0) __hip_atomic_fetch_add((int *)S.scal, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (merge_bc) gmm_dfeat();
                while (*(volatile int *)S.scal < 4 * (it + 1)) __builtin_amdgcn_s_sleep(1);
                BF_WAVE_FENCE();
                gmm_prior_grad_share();
            }'''
    assert old in s
    s = s.replace(old, new)
    # 2. wave 3 no longer forms the priors' gradient terms
    old = "        if (wave == 3 && GBLEND) gmm_prior_grad();       // (all eight q values are in LDS since the barrier behind phase D)"
    assert old in s
    s = s.replace(old, "        // (the priors' gradient terms: the GMM waves' phase F, gmm_prior_grad_share)")
    # 3. the share lambda next to gmm_prior_grad
    old = "    // phase B when the GMM waves blend: T_s = sum_j w_sj A_j of the wave's three vertices"
    new = '''    // the same, dealt over the four GMM waves (18 body dofs each): every wave takes the arg-min of the eight q itself
    auto gmm_prior_grad_share = [&]() {
        const int lq = bf_launder(lane);
        float gq[BF_GMM_M];
#pragma unroll
        for (int m = 0; m < BF_GMM_M; ++m) gq[m] = S.gq[m];
        int ms = 0;
        float qm = gq[0];
#pragma unroll
        for (int m = 1; m < BF_GMM_M; ++m) if (gq[m] < qm) { qm = gq[m]; ms = m; }
        if (lq == 0 && gwi == 0) { S.scal[1] = (float)ms; S.scal[2] = qm; }
        constexpr int PER = (BF_GMM_D + 3) / 4;
        const int pb = gwi * PER + lq;
        if (lq < PER && pb < T.nbp && pb < BF_GMM_D) {
            const float gyv = S.gy[ms * BF_GMM_LD + pb], pv = Pcur[T.off_pose + pb];
            const float sg = (pb == 52 ? 1.f : 0.f) - (pb == 55 ? 1.f : 0.f) - (pb == 9 ? 1.f : 0.f) - (pb == 12 ? 1.f : 0.f);
            const float ex = __expf(pv * sg);
            S.gtail[pb] = hp.w_pose * gyv;
            S.gtail[BF_GMM_LD + pb] = hp.w_angle * 2.f * ex * ex * sg;
        }
    };
    // phase B when the GMM waves blend: T_s = sum_j w_sj A_j of the wave's three vertices'''
    assert old in s
    s = s.replace(old, new)
    a = s.index("    auto gmm_prior_grad = [&]() {")
    b = s.index("    // the same, dealt over the four GMM waves (18 body dofs each)")
    s = s[:a] + s[b:]
    # 4. the counter starts at zero
    old = "    if (EXT && tid == 0) ((int *)S.part)[BF_POSE_STATE_FLAG] = 0;          // (the chain waves' cue: no iteration's token yet)"
    assert old in s
    s = s.replace(old, old + "\n    if (tid == 0) *(volatile int *)S.scal = 0;                               // (the GMM waves' join counter of phase F)")
    return s
