"""Deterministic synthetic checkpoints and PCM for parity tests and bench.py.

The reference's pretrained checkpoint is not obtainable offline (README.md:22 of the
reference points at a Google-Drive link), so parity and timing use seeded synthetic
weights written in the *checkpoint* layout the reference loads
(saber/trainer/manager/checkpoints.py:10-48: ``{"state": state_dict}``; weight-normed
layers stored as ``weight_g`` / ``weight_v``; key list: SURVEY.md App. A.4).

Pure numpy (``RandomState``), fixed key order, no torch RNG: the same bytes are
produced in the build container (where they are loaded into the reference to make the
golden fixtures) and on the GPU box.
"""
import numpy as np

N_TRI = 9976          # FLAME triangles (output_dim_scale 59856 = 9976*6, rotat 29928 = 9976*3)
N_VERT3 = 15069       # offsets head: 5023 vertices * 3
P = "_model."


def _bn(rs, sd, key, n):
    sd[key + ".weight"] = rs.uniform(0.8, 1.2, n).astype(np.float32)
    sd[key + ".bias"] = rs.normal(0, 0.1, n).astype(np.float32)
    sd[key + ".running_mean"] = rs.normal(0, 0.1, n).astype(np.float32)
    sd[key + ".running_var"] = rs.uniform(0.5, 1.5, n).astype(np.float32)
    sd[key + ".num_batches_tracked"] = np.asarray(1000, dtype=np.int64)


def _wn(rs, sd, key, shape, gain=2.0, bias=True):
    """weight-normed layer: weight = g * v / ||v|| (norm over all dims but 0)."""
    fan_in = int(np.prod(shape[1:]))
    v = rs.normal(0, np.sqrt(gain / fan_in), shape).astype(np.float32)
    norm = np.sqrt((v.astype(np.float64) ** 2).reshape(shape[0], -1).sum(1))
    g = (norm * rs.uniform(0.8, 1.2, shape[0])).astype(np.float32)
    if bias:
        sd[key + ".bias"] = rs.normal(0, 0.05, shape[0]).astype(np.float32)
    sd[key + ".weight_g"] = g.reshape((shape[0],) + (1,) * (len(shape) - 1))
    sd[key + ".weight_v"] = v


def _lstm(rs, sd, key, inp, hid, layer, bias, gain=1.0):
    k = gain / np.sqrt(hid)
    for suf in ("", "_reverse"):
        sd[f"{key}.weight_ih_l{layer}{suf}"] = rs.uniform(-k, k, (4 * hid, inp)).astype(np.float32)
        sd[f"{key}.weight_hh_l{layer}{suf}"] = rs.uniform(-k, k, (4 * hid, hid)).astype(np.float32)
        if bias:
            sd[f"{key}.bias_ih_l{layer}{suf}"] = rs.uniform(-k, k, 4 * hid).astype(np.float32)
            sd[f"{key}.bias_hh_l{layer}{suf}"] = rs.uniform(-k, k, 4 * hid).astype(np.float32)


def make_state_dict(head="dgrad", seed=1234):
    """Synthetic reference-layout state dict ({name: np.ndarray}), insertion-ordered."""
    assert head in ("dgrad", "offsets")
    rs = np.random.RandomState(seed)
    sd = {}
    enc = P + "_audio_encoder._layers."
    for idx, shape in ((1, (32, 3, 3, 1)), (3, (64, 32, 3, 1)), (5, (64, 64, 1, 1))):
        _wn(rs, sd, f"{enc}{idx}", shape)
        _bn(rs, sd, f"{enc}{idx}._ext_post_bn", shape[0])
    _lstm(rs, sd, f"{enc}6._lstm", 64, 128, 0, True, gain=1.5)
    sd[f"{enc}6._proj.weight"] = rs.normal(0, np.sqrt(8.0 / 8192), (256, 8192)).astype(np.float32)
    sd[f"{enc}6._proj.bias"] = rs.normal(0, 0.05, 256).astype(np.float32)
    _lstm(rs, sd, f"{enc}9", 256, 256, 0, False, gain=2.5)
    _lstm(rs, sd, f"{enc}9", 512, 256, 1, False, gain=2.5)
    sd[f"{enc}10.b"] = rs.normal(0, 0.1, (1, 1, 128)).astype(np.float32)
    sd[f"{enc}10._conv_query.weight"] = rs.normal(0, np.sqrt(4.0 / 1536), (512, 512, 3)).astype(np.float32)
    sd[f"{enc}10.proj_key.weight"] = rs.normal(0, np.sqrt(16.0 / 640), (128, 512)).astype(np.float32)
    sd[f"{enc}10.proj_qry.weight"] = rs.normal(0, np.sqrt(16.0 / 640), (128, 512)).astype(np.float32)
    sd[f"{enc}10.v.weight"] = rs.normal(0, 0.5, (1, 128)).astype(np.float32)
    out = P + "_output_module."
    if head == "dgrad":
        _wn(rs, sd, out + "_layers.0", (512, 520))
        for br, nc in (("_scale", 85), ("_rotat", 180)):
            _wn(rs, sd, f"{out}{br}_layers.0", (512, 520))
            _wn(rs, sd, f"{out}{br}_layers.1", (256, 512), gain=1.0)
            _wn(rs, sd, f"{out}{br}_layers.2", (nc, 256), gain=1.0)
        for br, nc, per in (("_scale", 85, 6), ("_rotat", 180, 3)):
            sd[f"{out}{br}_pca.compT"] = rs.normal(0, 0.02, (N_TRI * per, nc)).astype(np.float32)
            sd[f"{out}{br}_pca.means"] = rs.normal(0, 0.01, N_TRI * per).astype(np.float32)
    else:
        _wn(rs, sd, out + "_layers.0", (512, 520))
        _wn(rs, sd, out + "_layers.1", (256, 512), gain=1.0)
        _wn(rs, sd, out + "_layers.2", (59, 256), gain=1.0)
        sd[out + "_pca.compT"] = rs.normal(0, 0.02, (N_VERT3, 59)).astype(np.float32)
        sd[out + "_pca.means"] = rs.normal(0, 0.01, N_VERT3).astype(np.float32)
    return sd


def make_pcm(clip_index, n_samples, kind="uniform"):
    """Synthetic PCM in [-1, 1]: seed = 1000 + clip_index."""
    rs = np.random.RandomState(1000 + int(clip_index))
    if kind == "uniform":
        return rs.uniform(-0.5, 0.5, n_samples).astype(np.float32)
    if kind == "zeros":
        return np.zeros(n_samples, np.float32)
    if kind == "sweep":          # linear sine sweep 50 Hz .. 3.5 kHz at a nominal 16 kHz
        t = np.arange(n_samples, dtype=np.float64) / 16000.0
        dur = max(n_samples / 16000.0, 1e-9)
        ph = 2 * np.pi * (50.0 * t + 0.5 * (3450.0 / dur) * t * t)
        return (0.4 * np.sin(ph)).astype(np.float32)
    if kind == "speechlike":     # amplitude-modulated band noise: exercises the clamp floor and ceiling
        x = rs.uniform(-1, 1, n_samples)
        env = 0.5 * (1 + np.sin(2 * np.pi * 3.0 * np.arange(n_samples) / 16000.0)) ** 2
        y = np.convolve(x, np.ones(8) / 8, mode="same") * env * 0.9
        return np.clip(y, -0.999, 0.999).astype(np.float32)
    raise ValueError(kind)


def make_template_mesh(seed=7):
    """Synthetic template of FLAME's size for timing the mesh stage (the licensed FLAME geometry is not shipped): an open
    117 x 44 grid = 5148 vertices, 9976 triangles (the model's triangle count), 3887 vertices constrained -> 1261 free, FLAME's
    free-vertex count with the reference's non_face mask (speech_anime/datasets/vocaset/mask/non_face.py).
    Returns (verts float32 (V, 3), faces uint32 (9976, 3), constraint indices uint32)."""
    nx, ny = 117, 44
    rs = np.random.RandomState(seed)
    x, y = np.meshgrid(np.arange(nx) * 0.0015, np.arange(ny) * 0.003, indexing="ij")
    V = np.stack([x, y, 0.02 * np.sin(20 * x) * np.cos(15 * y)], -1).reshape(-1, 3).astype(np.float32)
    V += rs.normal(0, 1e-4, V.shape).astype(np.float32)
    idx = lambda i, j: i * ny + j
    F = np.asarray([[idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)] for i in range(nx - 1) for j in range(ny - 1)] +
                   [[idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)] for i in range(nx - 1) for j in range(ny - 1)], np.uint32)
    assert len(F) == N_TRI
    n_free = 1261
    cn = np.sort(rs.choice(len(V), len(V) - n_free, replace=False)).astype(np.uint32)
    return V, F, cn
