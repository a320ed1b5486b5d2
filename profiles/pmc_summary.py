#!/usr/bin/env python3
"""Per-kernel, per-launch-shape table from the three separate rocprofv3 --pmc passes (MfmaUtil, FETCH_SIZE, WRITE_SIZE).

Usage: python profiles/pmc_summary.py <dir with MfmaUtil_/FETCH_SIZE_/WRITE_SIZE_counter_collection.csv> [--traffic-json out.json]
                                      [--chunks 8192,8192,3968] [--frontend-json out.json --frames 20352] [--attention-json out.json]

--attention-json: the attention stage (north_star: "MFMA utilisation on the attention stage against gfx950 peak") = every dispatch
between a launch group's second BiLSTM recurrence and its attn_kernel (or attn_fused_f32_kernel, which is the whole layer in one
launch), inclusive: query Conv1d, query projection, key projection + scores, softmax / context.  MfmaUtil is weighted by each dispatch's duration in the same pass.

--chunks: the frames of the consecutive launch groups of one step (bench.py --chunk: 20,352 frames = 8192 + 8192 + 3968).  A
PERSISTENT kernel launches the same grid whatever the problem size, so (symbol, grid) cannot tell its 8192-frame launches from
its 3968-frame one (the round-2 table averaged them).  With --chunks the dispatches of every constant-grid symbol are
taken in dispatch order and split into len(chunks) consecutive groups -- the collection runs ONE step with no warm-up, so the
k-th group IS chunk k -- and printed per chunk in a second table.

Counter handling follows MI355X_MICROARCH.md, section HBM:
  * rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB on gfx950;
  * FETCH_SIZE counts a 128-byte request as 64 bytes for wide (16 B per lane) streaming reads -- which is how every
    kernel here reads its operands -- so the read column is DOUBLED ("read GB" = 2 x FETCH_SIZE);
  * WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
Rows are keyed by (kernel symbol, grid size).  Collect the passes with `bench.py --no-column-sharing
--no-mixed-precision`: a column-sharing run launches the SAME symbol and grid for the projection GEMMs but most of its
workgroups exit at once (q_limit), and the two would be averaged into one row (the round-1 summary did that).

--traffic-json writes the frequency-LSTM kernel's figures of the largest launch (a hardware-dispatched form: one workgroup per tile), which bench.py scales to its frames per launch
for `roofline.traffic`.
"""
import collections
import csv
import json
import os
import sys

FETCH_CORRECTION = 2.0      # MI355X_MICROARCH.md: "double it before comparing with a byte count"


def file_sha1(name):
    """sha1 of a kernel source as it was when the counters were collected (the script runs in the same snapshot): bench.py nulls
    a figure derived from these counters once the kernel file has changed."""
    import hashlib
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sdfa-2019_amd", "csrc", name)
    try:
        return hashlib.sha1(open(path, "rb").read()).hexdigest()
    except OSError:
        return None


def load(path):
    out = collections.defaultdict(list)
    for r in sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"])):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out[(name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return out


PERSISTENT = ("gemm_fat_kernel", "pca_dgrad_res_kernel", "freq_lstm_v3_kernel<false, true", "freq_lstm_v3_kernel<true, true",
              "freq_lstm_v2_kernel<false, true", "freq_lstm_v2_kernel<true, true")


def per_chunk(M, F, W, chunks):
    """Second table: dispatches of the constant-grid (persistent) kernels split by launch group."""
    print()
    print(f"persistent kernels by launch group (--chunks {','.join(map(str, chunks))}): dispatch order split into {len(chunks)} consecutive groups")
    print(f"{'kernel':46s} {'frames':>7s} {'calls':>5s} {'MfmaUtil %':>10s} {'read GB (x2)':>13s} {'write GB':>10s}")
    for k in sorted(M):
        if not k[0].startswith(PERSISTENT):
            continue
        m, f, w = M[k], F.get(k, []), W.get(k, [])
        if len(m) % len(chunks) or len(f) != len(m) or len(w) != len(m):
            print(f"{k[0][:46]:46s} -- {len(m)} dispatches do not split into {len(chunks)} groups (or the passes differ): skipped")
            continue
        per = len(m) // len(chunks)
        for ci, frames in enumerate(chunks):
            sl = slice(ci * per, (ci + 1) * per)
            mm, ff, ww = m[sl], f[sl], w[sl]
            print(f"{k[0][:46]:46s} {frames:7d} {per:5d} {sum(mm) / per:10.1f} {sum(ff) / per * 1024 * FETCH_CORRECTION / 1e9:13.3f} {sum(ww) / per * 1024 / 1e9:10.3f}")


def attention_stage(path):
    """Dispatches of the attention stage in the MfmaUtil pass, by position: after the second time_lstm* launch of a launch group, up
    to and including attn_kernel.  Returns (time-weighted MfmaUtil %, total ns, per-symbol rows)."""
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    n_lstm, inside, picked = 0, False, []
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if name.startswith("time_lstm"):
            n_lstm += 1
            inside = n_lstm % 2 == 0
            continue
        if inside and "at::" not in name and not name.startswith("__amd"):
            picked.append((name, float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
            if name.startswith(("attn_kernel", "attn_fused_f32_kernel")):      # the stage's last dispatch (round 6: the whole layer may be ONE launch)
                inside = False
    tot = sum(t for _, _, t in picked)
    by = collections.OrderedDict()
    for name, u, t in picked:
        e = by.setdefault(name, [0, 0.0, 0])
        e[0] += 1; e[1] += u * t; e[2] += t
    per = [{"kernel": k, "calls": v[0], "mfma_util_pct": round(v[1] / v[2], 2), "ns": v[2]} for k, v in by.items()]
    return (sum(u * t for _, u, t in picked) / tot if tot else None), tot, per


def attention_json(csv_path, out_path, label, mode="fp32"):
    """Writes the attention-stage record bench.py reads (attention_counter_util): the whole stage time-weighted, and the stage's MFMA
    kernels alone (everything but the softmax / context tail attn_kernel) -- VERDICT r5 asks for the latter in the bf16 attention mode."""
    util, ns, per = attention_stage(csv_path)
    if util is None:
        return None
    gem = [e for e in per if not e["kernel"].startswith("attn_kernel")]      # (attn_fused_f32_kernel IS a matrix kernel: it stays in)
    gns = sum(e["ns"] for e in gem)
    gutil = sum(e["mfma_util_pct"] * e["ns"] for e in gem) / gns if gns else None
    print()
    print(f"attention stage, precision {mode} (after the second BiLSTM recurrence .. attn_kernel), one step: {ns / 1e6:.3f} ms under the counter pass, "
          f"time-weighted MfmaUtil {util:.1f} % (its MFMA kernels alone: {gutil:.1f} % over {gns / 1e6:.3f} ms)")
    for e in per:
        print(f"  {e['kernel'][:60]:60s} calls {e['calls']:3d}  MfmaUtil {e['mfma_util_pct']:5.1f} %  {e['ns'] / 1e6:7.3f} ms")
    json.dump({"precision": mode, "mfma_util_pct_time_weighted": round(util, 2), "stage_ns_under_counters": ns,
               "mfma_util_pct_gemms": None if gutil is None else round(gutil, 2), "gemms_ns_under_counters": gns, "kernels": per,
               "attn_hip_sha1": file_sha1("attn.hip"), "gemm_hip_sha1": file_sha1("gemm.hip"),
               "source": f"rocprofv3 --pmc MfmaUtil pass ({label}), one step of the headline workload in precision {mode}; every dispatch between a launch "
                         "group's second time_lstm launch and its attn_kernel, weighted by its duration in that pass; mfma_util_pct_gemms leaves "
                         "out attn_kernel (softmax + context: vector work and the H stream, no matrix instructions)"},
              open(out_path, "w"), indent=1)
    return util


def main():
    if "--attention-only" in sys.argv:      # pmc_summary.py --attention-only <MfmaUtil csv> <out json> <precision mode> [label]
        i = sys.argv.index("--attention-only")
        csv_path, out, mode = sys.argv[i + 1:i + 4]
        attention_json(csv_path, out, sys.argv[i + 4] if len(sys.argv) > i + 4 else os.path.basename(os.path.dirname(os.path.abspath(csv_path))), mode)
        return
    d = sys.argv[1]
    M, F, W = (load(os.path.join(d, f"{c}_counter_collection.csv")) for c in ("MfmaUtil", "FETCH_SIZE", "WRITE_SIZE"))
    print(f"{'kernel':46s} {'grid':>10s} {'calls':>5s} {'MfmaUtil %':>10s} {'FETCH KiB raw':>14s} {'read GB (x2)':>13s} {'write GB':>10s}")
    for k in sorted(M, key=lambda k: -(sum(F.get(k, [0])) + sum(W.get(k, [0])))):
        f, w, m = F.get(k, [0]), W.get(k, [0]), M[k]
        if sum(f) + sum(w) < 1e5 or "at::" in k[0]:
            continue
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        print(f"{k[0][:46]:46s} {k[1]:10d} {len(m):5d} {sum(m) / len(m):10.1f} {fk:14.0f} {fk * 1024 * FETCH_CORRECTION / 1e9:13.3f} {wk * 1024 / 1e9:10.3f}")
    if "--chunks" in sys.argv:
        per_chunk(M, F, W, [int(x) for x in sys.argv[sys.argv.index("--chunks") + 1].split(",")])
    if "--frontend-json" in sys.argv:
        # the spectrogram stage (north_star: "rocprof counters reporting achieved HBM GB/s on the spectrogram stage"): counter bytes
        # of its two kernels per frame; bench.py divides by its live stage time
        frames = int(sys.argv[sys.argv.index("--frames") + 1])
        # round 5: the stage is share_prev_kernel + mel_stream_kernel (no mel table); with option frontend_two_kernel = 1 it is the share
        # map + mel_columns_kernel + gather_features_kernel of rounds 2-4.  Whichever form ran is summarised.
        fe = {}
        for k in M:
            for stem in ("mel_stream_repair", "mel_stream", "share_prev", "mel_columns", "gather_features"):      # first match: the repair pass is not the stream kernel
                if k[0].startswith(stem) and F.get(k) is not None and W.get(k) is not None and len(F[k]) and len(W[k]):
                    fe[stem] = {"read_bytes": F[k][-1] * 1024 * FETCH_CORRECTION, "write_bytes": W[k][-1] * 1024, "grid": k[1]}
                    break
        form = "stream" if "mel_stream" in fe else ("two_kernel" if len([s_ for s_ in ("mel_columns", "gather_features") if s_ in fe]) == 2 else None)
        if form == "stream":
            fe = {k: v for k, v in fe.items() if k in ("mel_stream", "mel_stream_repair", "share_prev")}      # the repair pass is part of the stage (one load per workgroup when healthy)
        elif form == "two_kernel":
            fe = {k: v for k, v in fe.items() if k in ("mel_columns", "gather_features")}
        if form:
            tot = sum(v["read_bytes"] + v["write_bytes"] for v in fe.values())
            names = "share_prev_kernel + mel_stream_kernel + mel_stream_repair_kernel" if form == "stream" else "mel_columns_kernel + gather_features_kernel"
            json.dump({"frames": frames, "form": form, "kernels": fe, "bytes_per_frame": tot / frames, "algorithmic_bytes_per_frame": 4 * 16000 / 60 + 98304,
                       "fetch_correction": FETCH_CORRECTION, "frontend_hip_sha1": file_sha1("frontend.hip"),
                       "source": f"separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes ({os.path.basename(os.path.normpath(d))}), last launch of "
                                 f"{names} (the full batch); FETCH_SIZE doubled per MI355X_MICROARCH.md"},
                      open(sys.argv[sys.argv.index("--frontend-json") + 1], "w"), indent=1)
    if "--attention-json" in sys.argv:
        attention_json(os.path.join(d, "MfmaUtil_counter_collection.csv"), sys.argv[sys.argv.index("--attention-json") + 1], os.path.basename(os.path.normpath(d)))
    if "--traffic-json" in sys.argv:
        key = max((k for k in M if k[0].startswith(("freq_lstm_v3_kernel<false, false", "freq_lstm_v2_kernel<false, false", "freq_lstm_kernel<false"))),
                  key=lambda k: (k[0].startswith("freq_lstm_v3"), k[0].startswith("freq_lstm_v2"), k[1]))      # hardware-dispatched forms: grid = tiles
        frames = key[1] // 256 // 2          # grid = (frames * 64 columns / 64 per workgroup) * 2 directions * 256 threads
        fk, wk = sum(F[key]) / len(F[key]), sum(W[key]) / len(W[key])
        read_b, write_b = fk * 1024 * FETCH_CORRECTION, wk * 1024
        alg_once = frames * (512 * 1024 + 2 * 1024 * 1024)
        alg_two = frames * (2 * 512 * 1024 + 2 * 1024 * 1024)
        json.dump({"kernel": key[0], "frames": frames, "lstm_hip_sha1": file_sha1("lstm.hip"), "fetch_kib_raw": round(fk), "write_kib": round(wk),
                   "fetch_correction": FETCH_CORRECTION, "read_bytes": round(read_b), "write_bytes": round(write_b),
                   "traffic_bytes": round(read_b + write_b),
                   "source": f"separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes ({os.path.basename(os.path.normpath(d))}); FETCH_SIZE doubled per "
                             "MI355X_MICROARCH.md (gfx950 tallies 128-byte read requests at 64 bytes)",
                   "algorithmic_bytes": alg_once,
                   "ratio_to_algorithmic": round((read_b + write_b) / alg_once, 4),
                   "note": f"algorithmic = conv3 activations read once (512 KiB/frame) + hidden states written once (2 MiB/frame) = "
                           f"{alg_once / 1e9:.2f} GB; measured read {read_b / 1e9:.2f} GB + write {write_b / 1e9:.2f} GB.  The forward and the "
                           f"backward recurrence of a column tile each stream the tile's activations, in opposite row order, a "
                           f"workgroup lifetime apart: read twice = {alg_two / 1e9:.2f} GB is what this kernel structure moves "
                           f"(see DESIGN.md section 4)"},
                  open(sys.argv[sys.argv.index("--traffic-json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
