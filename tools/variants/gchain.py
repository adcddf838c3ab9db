def patch(s):
    old = '''            g0 += Pa4.x * q0 + Pb4.x * q1 + Pc4.x * q2;
            g1 += Pa4.y * q0 + Pb4.y * q1 + Pc4.y * q2;
            g2 += Pa4.z * q0 + Pb4.z * q1 + Pc4.z * q2;'''
    new = '''            g0 = Pc4.x * q2 + (g0 + (Pa4.x * q0 + Pb4.x * q1));
            g1 = Pc4.y * q2 + (g1 + (Pa4.y * q0 + Pb4.y * q1));
            g2 = Pc4.z * q2 + (g2 + (Pa4.z * q0 + Pb4.z * q1));'''
    assert old in s
    return s.replace(old, new)
