"""In-situ A/B of the pipelined pointwise GEMM core (tools/experimental/pw_gemm/pw_gemm.hip; library built with AVDF_PW_EXPERIMENTAL=1) against the table plans: for each forward K = 1
shape of the exp10 step, re-capture the training step with the shape's plan row replaced by a pw configuration
(tile code 8 + cfg, workgroups per CU) and time the whole step.   python tools/experimental/pw_gemm/pw_insitu.py   (GPU box)"""
import ctypes, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import tools.plan_search as ps
import torch
from audio_visual_deepfake_detection_amd import build_model
from audio_visual_deepfake_detection_amd._lib import lib, check
from audio_visual_deepfake_detection_amd.engine import TrainStep
import copy, time

B = 8
M = B * 768
SHAPES = [(0, 1, 1, M, 1024, 256), (0, 1, 1, M, 256, 1024), (0, 1, 1, 3 * M, 256, 256), (0, 1, 1, M, 256, 256),
          (0, 1, 1, M // 2, 1024, 256), (0, 1, 1, M // 2, 256, 1024), (0, 1, 1, 3 * M // 2, 256, 256)]
CANDS = [(8 + c, 0) for c in (0, 1, 2, 4, 5, 6)]       # (plan code, workgroups per CU: 0 = what the LDS allows)


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model0, _ = build_model("deepfake_exp10.yaml")

    def measure(rows, steps=40):
        txt = ";".join("%d,%d,%d,%d,%d,%d,%d,%d" % (*k, a, b) for k, (a, b) in rows.items())
        check(lib.avdf_tuned_extra(ctypes.c_char_p(txt.encode()) if txt else None), "tuned_extra")
        model = copy.deepcopy(model0).to(dev)
        eng = TrainStep(model, B, ps.STREAMS, lr=1e-4)
        g = torch.Generator().manual_seed(1)
        for b in range(B):
            for dst in eng.streams[b]:
                dst.copy_(torch.randn(dst.shape, generator=g))
            eng.ngt[b] = 0
        eng.gt.zero_()
        eng.gt[0, 0, 0], eng.gt[0, 0, 1], eng.ngt[0] = 100.0, 260.0, 1
        eng.capture()
        for _ in range(8):
            eng.step()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            for _ in range(steps):
                eng.step()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps)
        del eng, model
        torch.cuda.empty_cache()
        return B / best

    base = measure({})
    print(f"baseline {base:.1f} clips/s", flush=True)
    out = {"baseline": round(base, 1), "shapes": []}
    for key in SHAPES:
        res = {}
        for a, b in CANDS:
            res[(a, b)] = measure({key: (a, b)})
            print(f"  {key} pw cfg {a - 8} wpc {b}: {res[(a, b)]:.1f}", flush=True)
        (a, b), v = max(res.items(), key=lambda kv: kv[1])
        print(f"{key}: best pw cfg {a - 8} wpc {b}: {v:.1f} vs table {base:.1f}", flush=True)
        out["shapes"].append({"shape": list(key), "best": [a, b], "clips_per_s": round(v, 1),
                              "all": {f"{x},{y}": round(z, 1) for (x, y), z in res.items()}})
        json.dump(out, open(os.path.join(ROOT, "gpurun_out", "pw_insitu.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
