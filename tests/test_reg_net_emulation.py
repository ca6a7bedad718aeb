"""CPU emulation of the register-resident value network (crowdnav_amd/csrc/sarl_reg_kernel.h): the weight-stream packing
(4 x 4 transposed output slots, bias quads, paired output tiles, the joint state's k order) and the lane maps of
v_mfma_f32_16x16x4_f32, restated in numpy, must reproduce sarl.ValueNetwork.forward (crowd_nav/policy/sarl.py:28-65).
The device kernel is tested against torch on the GPU (tests/test_sarl.py); this test pins the LAYOUT ALGEBRA where it can be
read: accumulator register s of output tile t in lane l  ==  B operand of k-step 4 t + s of the next layer."""
import numpy as np

DEPTH = 4
(L_MLP1_0, L_MLP1_2, L_MLP2_0, L_MLP2_2, L_ATT0_G, L_ATT0_L, L_ATT_2, L_ATT_4, L_MLP3_0, L_MLP3_2, L_MLP3_4, L_MLP3_6,
 LAYERS) = range(13)


def cdiv(a, b):
    return (a + b - 1) // b


def shape(l, xks):  # (k-steps, output tiles, bias quad, paired) — reg_shape()
    return {L_MLP1_0: (xks, 10, 1, 0), L_MLP1_2: (38, 7, 1, 0), L_MLP2_0: (25, 7, 1, 0), L_MLP2_2: (25, 4, 1, 0),
            L_ATT0_G: (25, 7, 1, 1), L_ATT0_L: (25, 7, 0, 0), L_ATT_2: (25, 7, 1, 0), L_ATT_4: (25, 1, 1, 0),
            L_MLP3_0: (15, 10, 1, 1), L_MLP3_2: (38, 7, 1, 1), L_MLP3_4: (25, 7, 1, 1), L_MLP3_6: (25, 1, 1, 1)}[l]


def tile_quads(l, xks):
    ks, _, bias, _ = shape(l, xks)
    return bias + cdiv(ks, 4)


def qbase(l, xks):
    return sum(shape(i, xks)[1] * tile_quads(i, xks) for i in range(l))


def qpos(l, xks, mt, j):
    _, MT, _, paired = shape(l, xks)
    S = tile_quads(l, xks)
    if not paired or mt >= (MT // 2) * 2:
        return mt * S + j
    return (mt // 2) * 2 * S + 2 * j + (mt & 1)


def kcol(l, K, k_off, ks, lg):
    if l != L_MLP3_0:
        return k_off + 4 * ks + lg if 4 * ks + lg < K else -1
    if ks < 12:
        return 6 + 4 * ks + lg
    if ks == 12:
        return 6 + 48 + lg if lg < 2 else lg
    if ks == 13:
        return 4 + lg if lg < 2 else -1
    return lg if lg < 2 else -1


def pack(params, xks):
    """params[l] = (W [N][ldw], b or None, K, k_off, replicate) -> stream [quads][64][4]"""
    total = cdiv(qbase(LAYERS, xks), DEPTH) * DEPTH
    stream = np.zeros((total, 64, 4), np.float32)
    for l in range(LAYERS):
        W, b, K, k_off, replicate = params[l]
        N = W.shape[0]
        KS, MT, bias, _ = shape(l, xks)
        for mt in range(MT):
            for j in range(tile_quads(l, xks)):
                quad = stream[qbase(l, xks) + qpos(l, xks, mt, j)]
                for lane in range(64):
                    lg, m = lane >> 4, lane & 15
                    for kk in range(4):
                        if bias and j == 0:
                            f = 0 if replicate else 16 * mt + 4 * kk + lg
                            quad[lane, kk] = b[f] if (b is not None and f < N) else 0.0
                        else:
                            ks = 4 * (j - bias) + kk
                            n = 0 if replicate else 16 * mt + 4 * (m & 3) + (m >> 2)
                            col = kcol(l, K, k_off, ks, lg) if ks < KS else -1
                            quad[lane, kk] = W[n, col] if (n < N and col >= 0) else 0.0
    return stream


def mfma(a, b, c):
    """v_mfma_f32_16x16x4_f32: a, b [64]; c [64][4].  A[i][k] = a[i + 16 k], B[k][j] = b[16 k + j]; lane l register s holds
    D[4 (l >> 4) + s][l & 15]."""
    A = a.reshape(4, 16).T            # [i][k]
    B = b.reshape(4, 16)              # [k][j]
    D = A.astype(np.float64) @ B.astype(np.float64)   # [i][j]
    d = c.astype(np.float64).copy()
    for l in range(64):
        for s in range(4):
            d[l, s] += D[4 * (l >> 4) + s, l & 15]
    return d.astype(np.float32)


class Stream:
    def __init__(self, stream, xks):
        self.s, self.xks, self.taken = stream, xks, []

    def take(self, l, mt, j):
        i = qbase(l, self.xks) + qpos(l, self.xks, mt, j)
        self.taken.append(i)
        return self.s[i]


def dense(ws, l, NT, relu, inp, init=None):
    """reg_dense / reg_dense1 (the pairing changes the stream order only): out[nt][mt] [64][4]"""
    KS, MT, bias, _ = shape(l, ws.xks)
    out = [[None] * MT for _ in range(NT)]
    order = []
    _, _, _, paired = shape(l, ws.xks)
    if paired:
        for p in range(MT // 2):
            order += [(2 * p, 2 * p + 1)]
        if MT & 1:
            order += [(MT - 1,)]
    else:
        order = [(mt,) for mt in range(MT)]
    for group in order:
        c0 = {mt: (ws.take(l, mt, 0) if bias else init(mt)) for mt in group}
        acc = {(nt, mt): None for nt in range(NT) for mt in group}
        for q in range(cdiv(KS, 4)):
            a = {mt: ws.take(l, mt, bias + q) for mt in group}
            for kk in range(4):
                ks = 4 * q + kk
                if ks < KS:
                    for mt in group:
                        for nt in range(NT):
                            c = c0[mt] if ks == 0 else acc[(nt, mt)]
                            acc[(nt, mt)] = mfma(a[mt][:, kk], inp(nt, ks), c)
        for mt in group:
            for nt in range(NT):
                out[nt][mt] = np.maximum(acc[(nt, mt)], 0) if relu else acc[(nt, mt)]
    return out


def reg_forward(stream, xks, X, cnt):
    """X [5][xks][64] (the feature kernel's fragment order), cnt [16] -> value [16]; mirrors sarl_reg_kernel."""
    NT = 5
    ws = Stream(stream, xks)
    lane = np.arange(64)
    c = cnt[lane & 15]
    reg = lambda T: (lambda nt, ks: T[nt][ks >> 2][:, ks & 3])
    h1 = dense(ws, L_MLP1_0, NT, True, lambda nt, ks: X[nt][ks])
    h2 = dense(ws, L_MLP1_2, NT, True, reg(h1))
    t1 = dense(ws, L_MLP2_0, NT, True, reg(h2))
    feat = dense(ws, L_MLP2_2, NT, False, reg(t1))
    gm = []
    for t in range(7):
        s = np.zeros((64, 4), np.float32)
        for nt in range(NT):
            s = s + np.where((nt < c)[:, None], h2[nt][t], 0).astype(np.float32)
        gm.append((s / c[:, None].astype(np.float32)).astype(np.float32))
    gterm = dense(ws, L_ATT0_G, 1, False, lambda nt, ks: gm[ks >> 2][:, ks & 3])[0]
    a0 = dense(ws, L_ATT0_L, NT, True, reg(h2), init=lambda mt: gterm[mt])
    att = dense(ws, L_ATT_2, NT, True, reg(a0))
    sc = dense(ws, L_ATT_4, NT, False, reg(att))
    e = []
    for nt in range(NT):
        s = sc[nt][0][:, 0]
        assert np.array_equal(s, sc[nt][0][:, 3])  # replicated over the 16 output slots
        e.append(np.where(nt < c, np.exp(s) * (s != 0), 0).astype(np.float32))
    total = sum(e)
    e = [x / total for x in e]
    wf = [sum(e[nt][:, None] * feat[nt][t] for nt in range(NT)).astype(np.float32) for t in range(4)]
    self0, self1 = X[0][0], X[0][1]
    mix = np.where(lane < 32, wf[3][:, 0], self0)
    joint = lambda nt, ks: wf[ks >> 2][:, ks & 3] if ks < 12 else mix if ks == 12 else self1 if ks == 13 else self0
    j1 = dense(ws, L_MLP3_0, 1, True, joint)[0]
    j2 = dense(ws, L_MLP3_2, 1, True, lambda nt, ks: j1[ks >> 2][:, ks & 3])[0]
    j3 = dense(ws, L_MLP3_4, 1, True, lambda nt, ks: j2[ks >> 2][:, ks & 3])[0]
    val = dense(ws, L_MLP3_6, 1, False, lambda nt, ks: j3[ks >> 2][:, ks & 3])[0]
    # every quad of the stream is consumed exactly once, in ascending order (the kernel's prefetch is a linear walk)
    assert ws.taken == list(range(qbase(LAYERS, xks)))
    return val[0][:16, 0]


def reference_forward(P, x, cnt):
    """sarl.py:28-65 for one group: x [H][in_dim], cnt humans present"""
    lin = lambda i, v: v @ P['W%d' % i].T.astype(np.float64) + P['b%d' % i]
    relu = lambda v: np.maximum(v, 0)
    x = x[:cnt].astype(np.float64)
    h2 = relu(lin(1, relu(lin(0, x))))
    feat = lin(3, relu(lin(2, h2)))
    g = np.broadcast_to(h2.mean(0, keepdims=True), h2.shape)
    a = relu(lin(5, relu(lin(4, np.concatenate([h2, g], 1)))))
    s = lin(6, a)[:, 0]
    w = np.exp(s) * (s != 0)
    w = w / w.sum()
    joint = np.concatenate([x[0, :6], (w[:, None] * feat).sum(0)])
    v = joint
    for i in (7, 8, 9):
        v = relu(lin(i, v))
    return lin(10, v)[0]


def _check(in_dim, seed):
    rng = np.random.default_rng(seed)
    xks = 4 if in_dim == 13 else 16
    dims = [(150, in_dim), (100, 150), (100, 100), (50, 100), (100, 200), (100, 100), (1, 100), (150, 56), (100, 150),
            (100, 100), (1, 100)]
    P = {}
    for i, (n, k) in enumerate(dims):
        P['W%d' % i] = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
        P['b%d' % i] = (0.1 * rng.standard_normal(n)).astype(np.float32)
    sd = [0, 1, 2, 3, 4, 4, 5, 6, 7, 8, 9, 10]
    params = []
    for l in range(LAYERS):
        W, b = P['W%d' % sd[l]], P['b%d' % sd[l]]
        K, k_off, rep = W.shape[1], 0, 0
        if l in (L_ATT0_G, L_ATT0_L):
            K = 100
        if l == L_ATT0_G:
            k_off = 100
        if l == L_ATT0_L:
            b = None
        if l == L_ATT_4:
            rep = 1
        params.append((W, b, K, k_off, rep))
    stream = pack(params, xks)
    x = rng.standard_normal((16, 5, in_dim)).astype(np.float32)      # [group][human][feature]
    x[:, 1:, :6] = x[:, :1, :6]                                      # the self part of a row is the same for every human
    cnt = np.array([5, 5, 4, 3, 5, 1, 2, 5, 5, 5, 4, 5, 3, 5, 5, 2])
    ks_x = 20 if xks == 16 else 5
    X = np.zeros((5, ks_x, 64), np.float32)                          # sarl_feature_kernel's order
    for h in range(5):
        for n in range(in_dim):
            X[h, n >> 2, (n & 3) * 16:(n & 3) * 16 + 16] = x[:, h, n]
    got = reg_forward(stream, xks, X, cnt)
    want = np.array([reference_forward(P, x[g], cnt[g]) for g in range(16)])
    assert np.abs(got - want).max() < 2e-5, np.abs(got - want).max()


def test_register_network_layout_plain():
    _check(13, 0)


def test_register_network_layout_with_occupancy_maps():
    _check(61, 1)


# ---- lstm_reg_kernel: the gate layer's packing and the cell update in the accumulator layout ---------------------------------
# Output tile mt holds gates i, f, g, o (torch order) of units 4 mt .. 4 mt + 3: accumulator register kk of lane group lg =
# gate kk of unit 4 mt + lg, so the new hidden state of a tile is, lane for lane, the B operand of k-step XKS + mt of the next
# human's gate layer; the head reads joint = [self (6) | h_n (50)] through the k order of its first layer.
HID, KSH = 50, 13


def pack_gates(Wih, Whh, bih, bhh, xks):
    K = Wih.shape[1]
    KS = xks + KSH
    S = 1 + cdiv(KS, 4)
    stream = np.zeros((KSH * S, 64, 4), np.float32)
    for mt in range(KSH):
        for j in range(S):
            quad = stream[mt * S + j]
            for lane in range(64):
                lg, m = lane >> 4, lane & 15
                for kk in range(4):
                    if j == 0:
                        u = 4 * mt + lg
                        quad[lane, kk] = bih[kk * HID + u] + bhh[kk * HID + u] if u < HID else 0.0
                    else:
                        ks, u = 4 * (j - 1) + kk, 4 * mt + (m >> 2)
                        n = (m & 3) * HID + u
                        if u < HID and ks < KS:
                            if ks < xks:
                                quad[lane, kk] = Wih[n, 4 * ks + lg] if 4 * ks + lg < K else 0.0
                            else:
                                c = 4 * (ks - xks) + lg
                                quad[lane, kk] = Whh[n, c] if c < HID else 0.0
    return stream, S


def test_lstm_gate_layer_layout_and_head_k_order():
    rng = np.random.default_rng(3)
    T, D, xks = 4, 13, 4
    Wih = (rng.standard_normal((4 * HID, D)) / np.sqrt(D)).astype(np.float32)
    Whh = (rng.standard_normal((4 * HID, HID)) / np.sqrt(HID)).astype(np.float32)
    bih, bhh = (0.1 * rng.standard_normal((2, 4 * HID))).astype(np.float32)
    W0 = (rng.standard_normal((150, 6 + HID)) / np.sqrt(56)).astype(np.float32)
    b0 = (0.1 * rng.standard_normal(150)).astype(np.float32)
    x = rng.standard_normal((16, T, D)).astype(np.float32)  # [group][human = LSTM step][feature]
    x[:, 1:, :6] = x[:, :1, :6]
    X = np.zeros((T, xks, 64), np.float32)
    for t in range(T):
        for n in range(D):
            X[t, n >> 2, (n & 3) * 16:(n & 3) * 16 + 16] = x[:, t, n]
    stream, S = pack_gates(Wih, Whh, bih, bhh, xks)
    sig = lambda v: 1.0 / (1.0 + np.exp(-v.astype(np.float64)))
    h = [np.zeros(64, np.float32) for _ in range(KSH)]
    c = [np.zeros(64, np.float32) for _ in range(KSH)]
    for t in range(T):
        hn = []
        for mt in range(KSH):
            acc = stream[mt * S].copy()
            for ks in range(xks + KSH):
                a = stream[mt * S + 1 + ks // 4][:, ks % 4]
                acc = mfma(a, X[t][ks] if ks < xks else h[ks - xks], acc)
            cn = sig(acc[:, 1]) * c[mt] + sig(acc[:, 0]) * np.tanh(acc[:, 2].astype(np.float64))
            c[mt] = cn.astype(np.float32)
            hn.append((sig(acc[:, 3]) * np.tanh(cn)).astype(np.float32))
        h = hn
    # numpy LSTM (torch.nn.LSTM's equations, gate order i, f, g, o)
    hr, cr = np.zeros((16, HID)), np.zeros((16, HID))
    for t in range(T):
        g = x[:, t].astype(np.float64) @ Wih.T + bih + hr @ Whh.T + bhh
        i, f, gg, o = (g[:, k * HID:(k + 1) * HID] for k in range(4))
        cr = sig(f) * cr + sig(i) * np.tanh(gg)
        hr = sig(o) * np.tanh(cr)
    lane = np.arange(64)
    for j in range(KSH):
        u = 4 * j + (lane >> 4)
        want = np.where(u < HID, hr[lane & 15, np.minimum(u, HID - 1)], 0.0)
        assert np.abs(h[j] - want).max() < 2e-6, (j, np.abs(h[j] - want).max())
    # head layer 0 on joint = [self | h_n]: k-steps 0..12 = h's registers, 13 = X k-step 0 of the first row, 14 = its k-step 1
    def kcol_head(ks, lg):
        if ks < KSH:
            return 6 + 4 * ks + lg if 4 * ks + lg < HID else -1
        return lg if ks == 13 else (4 + lg if lg < 2 else -1)
    inp = lambda ks: h[ks] if ks < KSH else (X[0][0] if ks == 13 else X[0][1])
    y = np.zeros((16, 160))
    for mt in range(10):
        acc = np.zeros((64, 4), np.float32)
        for l_ in range(64):
            for kk in range(4):
                f = 16 * mt + 4 * kk + (l_ >> 4)
                acc[l_, kk] = b0[f] if f < 150 else 0.0
        for ks in range(15):
            a = np.zeros(64, np.float32)
            for l_ in range(64):
                lg, m = l_ >> 4, l_ & 15
                n, col = 16 * mt + 4 * (m & 3) + (m >> 2), kcol_head(ks, lg)
                a[l_] = W0[n, col] if (n < 150 and col >= 0) else 0.0
            acc = mfma(a, inp(ks), acc)
        for l_ in range(64):
            for kk in range(4):
                y[l_ & 15, 16 * mt + 4 * kk + (l_ >> 4)] = acc[l_, kk]
    joint = np.concatenate([x[:, 0, :6].astype(np.float64), hr], axis=1)
    want = joint @ W0.T + b0
    assert np.abs(y[:, :150] - want).max() < 2e-5, np.abs(y[:, :150] - want).max()


def test_hoisted_occupancy_map_term_layout():
    """kRegSarlPre: mlp1.0's accumulators start from b + W[:, 13:61] om, computed once per (env, human) and stored in the
    accumulator's own order (sarl_om_term_kernel: term[row][16 mt + 4 lg + kk] = feature 16 mt + 4 kk + lg); 4 k-steps of the 13
    rotated features follow.  Must equal the 61-input layer."""
    rng = np.random.default_rng(5)
    W = (rng.standard_normal((150, 61)) / np.sqrt(61)).astype(np.float32)
    b = (0.1 * rng.standard_normal(150)).astype(np.float32)
    x = rng.standard_normal((16, 61)).astype(np.float32)  # one human of 16 groups (here: 16 different envs)
    term = np.zeros((16, 160), np.float32)                # [row][slot]
    for slot in range(160):
        mt, lg, kk = slot >> 4, (slot >> 2) & 3, slot & 3
        f = 16 * mt + 4 * kk + lg
        if f < 150:
            v = np.float32(b[f])
            for k in range(48):
                v = np.float32(np.float32(W[f, 13 + k] * x[:, 13 + k]) + v)  # fma chain, bias first (one rounding more than fma: 1e-6)
            term[:, slot] = v
    X = np.zeros((4, 64), np.float32)
    for n in range(13):
        X[n >> 2, (n & 3) * 16:(n & 3) * 16 + 16] = x[:, n]
    lane = np.arange(64)
    got = np.zeros((16, 160))
    for mt in range(10):
        acc = np.zeros((64, 4), np.float32)
        for kk in range(4):
            acc[:, kk] = term[lane & 15, mt * 16 + (lane >> 4) * 4 + kk]  # one 16-byte load per lane
        for ks in range(4):
            a = np.zeros(64, np.float32)
            for l_ in range(64):
                lg, m = l_ >> 4, l_ & 15
                n, col = 16 * mt + 4 * (m & 3) + (m >> 2), 4 * ks + lg
                a[l_] = W[n, col] if (n < 150 and col < 13) else 0.0
            acc = mfma(a, X[ks], acc)
        for l_ in range(64):
            for kk in range(4):
                got[l_ & 15, 16 * mt + 4 * kk + (l_ >> 4)] = acc[l_, kk]
    want = x.astype(np.float64) @ W.T.astype(np.float64) + b
    assert np.abs(got[:, :150] - want).max() < 1e-5, np.abs(got[:, :150] - want).max()
