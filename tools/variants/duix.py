def patch(s):
    old = '''            const v2f du = kk * rx * ix * ix, dw = kk * ry * iy * iy;'''
    new = '''            const v2f du = (kk * ix) * (rx * ix), dw = (kk * iy) * (ry * iy);'''
    assert old in s
    return s.replace(old, new)
