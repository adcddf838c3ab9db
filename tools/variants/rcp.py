def patch(s):
    old = '''            const v2f ix = {__builtin_amdgcn_rcpf(dx.x), __builtin_amdgcn_rcpf(dx.y)};
            const v2f iy = {__builtin_amdgcn_rcpf(dy.x), __builtin_amdgcn_rcpf(dy.y)};'''
    new = '''            const v2f dxy = dx * dy;
            const v2f ixy = {__builtin_amdgcn_rcpf(dxy.x), __builtin_amdgcn_rcpf(dxy.y)};
            const v2f ix = ixy * dy, iy = ixy * dx;'''
    assert old in s
    return s.replace(old, new)
