"""Mixed-precision tolerance sweep (BASELINE configs[3]): what dgrad error a split-bf16 MFMA path would have.

Runs the golden model inputs through analysis builds of the library in which EVERY MFMA operand (weights and
activations of all GEMMs, convs and LSTM recurrences) is cut to its n leading bfloat16 terms (csrc: make TERMS=n),
while accumulation stays fp32 and the mel front end / softmax stay fp32.  n = 1 ~ plain bf16 operands,
n = 2 ~ bf16x3 (hi*hi + hi*lo + lo*hi), n = 3 ~ bf16x6 (fp32-equivalent).  Prints one JSON object.
Usage (GPU box):  python tools/precision_sweep.py            (spawns one subprocess per build)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sdfa-2019_amd")

CHILD = r"""
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
from sdfa_amd import synth
from sdfa_amd.engine import Engine
g = np.load(%r)
e = np.load(%r)
sd = synth.make_state_dict("dgrad", 1234)
eng = Engine(sd)
x = torch.from_numpy(g["audio_feat"]).cuda()
out, z, align, coef = eng.forward(x, torch.full((x.shape[0],), 2, dtype=torch.int64), want_coef=True)
out = out.cpu().numpy()
res = dict(dgrad=float(np.abs(out[:, ::97] - g["dgrad_stride97"]).max()), z=float(np.abs(z.cpu().numpy() - g["z"][:, 0]).max()),
           align=float(np.abs(align.cpu().numpy() - g["align"][:, 0]).max()),
           coef=float(np.abs(coef.cpu().numpy() - np.concatenate([g["coef_scale"][:, 0], g["coef_rotat"][:, 0]], 1)).max()))
feat, ts, cnt = eng.mel_frontend([synth.make_pcm(0, 32000)], 16000)
o2, *_ = eng.forward(feat, torch.full((feat.shape[0],), 2, dtype=torch.int64))
o2 = o2.cpu().numpy().reshape(feat.shape[0], 9976, 9)
res["dgrad_e2e_2s_clip"] = float(np.abs(o2[:, ::97] - e["sr16000_stride97"]).max())
print("RESULT " + json.dumps(res))
"""


def main():
    table = {}
    for name, lib in (("fp32 (product)", "libsdfa_hip.so"), ("3 bf16 terms", "libsdfa_hip_terms3.so"),
                      ("2 bf16 terms", "libsdfa_hip_terms2.so"), ("1 bf16 term", "libsdfa_hip_terms1.so")):
        path = os.path.join(PKG, "sdfa_amd", lib)
        if not os.path.exists(path):
            table[name] = "not built (make -C sdfa-2019_amd/csrc TERMS=n)"
            continue
        env = dict(os.environ, SDFA_HIP_LIB=path)
        code = CHILD % (PKG, os.path.join(ROOT, "tests", "golden", "model_dgrad.npz"), os.path.join(ROOT, "tests", "golden", "e2e_dgrad.npz"))
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        table[name] = json.loads(line[0][7:]) if line else {"error": p.stderr[-400:]}
    print(json.dumps({"reference": "tests/golden (reference PyTorch CPU path, fp32)", "tolerance_north_star": 1e-4,
                      "max_abs_error": table}, indent=1))


if __name__ == "__main__":
    main()
