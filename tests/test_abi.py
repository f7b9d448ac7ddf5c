"""CPU checks of the C-ABI boundary: the shared library loads, exports every symbol that
include/avdf.h declares, and the ctypes signatures in _lib.py agree with the header."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _parse_header():
    src = open(os.path.join(ROOT, "include", "avdf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(avdf_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        if args == "void":
            kinds = []
        else:
            kinds = []
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    kinds.append("P")
                elif a.startswith("float"):
                    kinds.append("F")
                elif a.startswith("int64_t"):
                    kinds.append("L")
                elif a.startswith("int"):
                    kinds.append("I")
                else:
                    raise AssertionError(f"unparsed arg {a!r} in {name}")
        decls[name] = (ret, kinds)
    return decls


def test_header_matches_ctypes_signatures():
    from audio_visual_deepfake_detection_amd import _lib
    decls = _parse_header()
    assert set(decls) == set(_lib.SIGNATURES), set(decls) ^ set(_lib.SIGNATURES)
    cmap = {ctypes.c_void_p: "P", ctypes.c_int: "I", ctypes.c_float: "F", ctypes.c_int64: "L"}
    rmap = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "const char*": ctypes.c_char_p}
    for name, (ret, kinds) in decls.items():
        res, args = _lib.SIGNATURES[name]
        assert res is rmap[ret], name
        assert [cmap[a] for a in args] == kinds, (name, [cmap[a] for a in args], kinds)


def test_library_exports_every_declared_symbol():
    from audio_visual_deepfake_detection_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _parse_header():
        assert hasattr(lib, name), name
    assert _lib.lib.avdf_version() >= 1


def test_host_nms_matches_reference_vectors(golden_dir):
    import numpy as np
    from audio_visual_deepfake_detection_amd import _lib
    g = np.load(os.path.join(golden_dir, "nms.npz"))
    for i in range(int(g["n_cases"])):
        segs = np.ascontiguousarray(g[f"segs{i}"], dtype=np.float32)
        sc = np.ascontiguousarray(g[f"scores{i}"], dtype=np.float32)
        n = sc.shape[0]
        keep = np.zeros(max(n, 1), dtype=np.int64)
        nk = ctypes.c_int64(0)
        rc = _lib.lib.avdf_nms1d_host(segs.ctypes.data if n else None, sc.ctypes.data if n else None, n, 0.1,
                                      keep.ctypes.data, ctypes.addressof(nk))
        assert rc == 0
        ref = g[f"keep{i}"]
        got = keep[:nk.value]
        # the reference's sort is unstable (nms_cpu.cpp:28): among EXACT score ties its order is
        # implementation-defined, so compare the kept set and the descending score sequence
        assert sorted(got.tolist()) == sorted(ref.tolist()), i
        assert np.array_equal(sc[got], sc[ref]), i


def test_batched_nms_api_matches_reference_outputs(golden_dir):
    """libs.utils.batched_nms (reference API) vs the reference's own outputs incl. voting, two min_score settings."""
    import numpy as np
    import torch
    from audio_visual_deepfake_detection_amd.libs.utils import batched_nms
    g = np.load(os.path.join(golden_dir, "nms.npz"))
    for i in range(int(g["n_cases"])):
        segs, sc = torch.from_numpy(g[f"segs{i}"]).reshape(-1, 2), torch.from_numpy(g[f"scores{i}"])
        for tag, ms in (("a", 0.001), ("b", 0.2)):
            s, c, l = batched_nms(segs, sc, torch.zeros(sc.shape[0], dtype=torch.int64), 0.1, ms, 100,
                                  use_soft_nms=False, multiclass=False, sigma=0.75, voting_thresh=0.9)
            rs, rc = g[f"bn_segs{i}{tag}"], g[f"bn_scores{i}{tag}"]
            assert s.shape[0] == rs.shape[0], (i, tag)
            assert np.array_equal(np.sort(c.numpy())[::-1], np.sort(rc)[::-1])
            # order-insensitive among exact score ties (the reference sort is unstable)
            o1 = np.lexsort((s.numpy()[:, 0], -c.numpy()))
            o2 = np.lexsort((rs[:, 0], -rc))
            assert np.allclose(s.numpy()[o1], rs[o2], atol=1e-3), (i, tag)


def test_batched_nms_soft_and_multiclass_branches_match_reference(golden_dir):
    """(soft | hard) x (multiclass | agnostic) branches of libs.utils.batched_nms vs the reference's outputs
    (tests/golden/nms_modes.npz: its compiled nms_1d_cpu.nms / .softnms + seg_voting), incl. n = 0 / 1 and score ties."""
    import numpy as np
    import torch
    from audio_visual_deepfake_detection_amd.libs.utils import batched_nms
    g = np.load(os.path.join(golden_dir, "nms_modes.npz"))
    for i in range(int(g["n_cases"])):
        a = g[f"c{i}_in"].reshape(-1, 4)
        soft, multi = (bool(v) for v in g[f"c{i}_mode"])
        s, c, l = batched_nms(torch.from_numpy(a[:, :2].copy()), torch.from_numpy(a[:, 2].copy()),
                              torch.from_numpy(a[:, 3].astype(np.int64)), 0.1, 0.001, 100, use_soft_nms=soft, multiclass=multi,
                              sigma=0.5, voting_thresh=0.75)
        rs, rc, rl = g[f"c{i}_segs"], g[f"c{i}_scores"], g[f"c{i}_cls"]
        assert s.shape[0] == rs.shape[0], (i, soft, multi, s.shape, rs.shape)
        if rs.shape[0] == 0:
            continue
        assert np.allclose(np.sort(c.numpy())[::-1], np.sort(rc)[::-1], atol=1e-6), (i, soft, multi)
        o1 = np.lexsort((s.numpy()[:, 0], l.numpy(), -np.round(c.numpy(), 5)))
        o2 = np.lexsort((rs[:, 0], rl, -np.round(rc, 5)))
        assert np.allclose(s.numpy()[o1], rs[o2], atol=1e-3), (i, soft, multi)
        assert np.array_equal(l.numpy()[o1], rl[o2]), (i, soft, multi)


def test_reference_named_nms_entry_points(golden_dir):
    """The reference's names (libs/utils/nms.py:5,8,38): an importable `nms_1d_cpu` with nms / softnms, and the NMSop /
    SoftNMSop autograd-Function classes, against the reference extension's own outputs (tests/golden/nms.npz)."""
    import numpy as np
    import torch
    from audio_visual_deepfake_detection_amd.libs.utils import nms_1d_cpu
    from audio_visual_deepfake_detection_amd.libs.utils.nms import NMSop, SoftNMSop
    g = np.load(os.path.join(golden_dir, "nms.npz"))
    for i in range(int(g["n_cases"])):
        segs, sc = torch.from_numpy(g[f"segs{i}"]).reshape(-1, 2), torch.from_numpy(g[f"scores{i}"])
        inds = nms_1d_cpu.nms(segs, sc, iou_threshold=0.1)
        assert inds.dtype == torch.int64 and sorted(inds.tolist()) == sorted(g[f"keep{i}"].tolist())
        cls = torch.zeros(sc.shape[0], dtype=torch.int64)
        s, c, l = NMSop.apply(segs, sc, cls, 0.1, 0.0, 3)
        assert s.shape[0] == min(3, inds.shape[0]) and torch.equal(c, sc[inds[:3]]) and torch.equal(s, segs[inds[:3]])
        # method 0 of the soft kernel IS hard suppression (suppressed scores become 0 and fall below min_score): same kept set
        dets = torch.zeros((sc.shape[0], 3))
        si = nms_1d_cpu.softnms(segs, sc, dets, iou_threshold=0.1, sigma=0.5, min_score=1e-9, method=0)
        assert sorted(si.tolist()) == sorted(inds.tolist())
        assert torch.equal(dets[:si.shape[0], 2], sc[si]) and torch.equal(dets[:si.shape[0], :2], segs[si])
        s2, c2, l2 = SoftNMSop.apply(segs, sc, cls, 0.1, 0.5, 0.001, 2, 100)
        assert s2.shape[0] == c2.shape[0] == l2.shape[0] and (c2.shape[0] == 0 or float(c2.max()) <= float(sc.max()) + 1e-6)
