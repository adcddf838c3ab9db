def patch(s):
    old = "    s.feat = take(npf);                s.vpp = take(BF_FIT_THREADS + ns * 3);"
    new = "    s.feat = take(npf + 8);            s.vpp = take(BF_FIT_THREADS + ns * 3);      // (+ 8 zeros behind the pose feature: the blend's last row slice runs past it)"
    assert old in s
    s = s.replace(old, new)
    old = '''                const int p = p0 + 2 * (h * PB + i);
                acc += (p < NPF ? f[i].x : 0.f) * w[i].x;
                acc += (p + 1 < NPF ? f[i].y : 0.f) * w[i].y;'''
    new = '''                acc += f[i].x * w[i].x;          // (rows past the end of the pose feature: zeros in the table AND behind the feature)
                acc += f[i].y * w[i].y;'''
    assert old in s
    s = s.replace(old, new)
    old = "        constexpr int NPF = NJ > 0 ? 9 * (NJ - 1) : 8;\n"
    assert old in s
    s = s.replace(old, "")
    old = "    if (EXT && tid == 0) ((int *)S.part)[BF_POSE_STATE_FLAG] = 0;          // (the chain waves' cue: no iteration's token yet)"
    assert old in s
    s = s.replace(old, old + "\n    if (tid < 8) S.feat[npf + tid] = 0.f;                                     // (the zeros behind the pose feature, see gmm_blend)")
    return s
