def patch(s):
    old = '''                    const float v = pr[r * 3] * rel0 + pr[r * 3 + 1] * rel1 + pr[r * 3 + 2] * rel2 + pt[r];'''
    new = '''                    float v = pr[r * 3] * rel0 + pt[r]; v = pr[r * 3 + 1] * rel1 + v; v = pr[r * 3 + 2] * rel2 + v;'''
    assert old in s
    return s.replace(old, new)
