#!/usr/bin/env python3
"""Generate tests/golden/*.json|*.npz from the reference tree.  CONTAINER ONLY.

Needs /root/reference (read-only) and oracle/_ref/libbt709ref.so (built by
`make -C oracle` from the reference headers where they lie).  Everything written
here is DATA -- numbers asserted by the reference's XCTest files, outputs of the
reference's own header functions, and small crops of images the reference
bundles, passed through the reference's encode functions -- never source text.

    python tests/golden/make_golden.py

Outputs (all committed):
  vectors.json      test-suite vectors parsed from EmptyiOSTests/*.m (file:line kept)
  reference.json    table hashes, histograms, threshold tables, alpha map, produced by
                    running the reference headers (oracle/_ref)
  patterns.npz      crops of the bundled test images as NV12 (reference encoder) with the
                    reference decode of each crop in every gamma mode
  patterns_full.npz WHOLE bundled test images as NV12 (reference encoder): the 1920x1080 QuickTime pattern in
                    both of its renditions and the 512x512 Image.tga; patterns_full.json holds the sha256 of
                    the reference headers' decode of each in every gamma mode, and of decode + 2:1 pass 2
                    (`python tests/golden/make_golden.py --full-patterns` regenerates only these two)
  pass2.json        sha256 of decode + pass 2 (2:1 and any-ratio rescale) composed of the reference's
                    own inlines (oracle/ref_harness.c) on seeded frames and on the pattern crops;
                    pass2_small.npz holds the small outputs themselves
"""
import hashlib
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_lib import Reference  # noqa: E402

REF = "/root/reference"
TESTS = os.path.join(REF, "EmptyiOSTests")


# ----------------------------------------------------------------- XCTest parsing

def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def methods(path):
    """Yield (name, first_line, body) for every `- (void)testXxx {` method."""
    src = strip_comments(open(path, encoding="utf-8", errors="replace").read())
    heads = [(m.start(), m.group(1)) for m in re.finditer(r"^- \(void\)\s*(test\w+)\s*\{", src, flags=re.M)]
    for i, (pos, name) in enumerate(heads):
        end = heads[i + 1][0] if i + 1 < len(heads) else len(src)
        yield name, src.count("\n", 0, pos) + 1, src[pos:end]


ASSIGN = re.compile(r"\b(?:int|uint32_t)?\s*\b(Rin|Gin|Bin)\s*=\s*([^;]+);")
EXPECT = re.compile(r"int\s+v\s*=\s*(\w+)\s*;\s*int\s+expectedVal\s*=\s*([^;]+);", re.S)
TYPE = re.compile(r"BGRAToBT709ConverterTypeEnum\s+(\w+)\s*=\s*BGRAToBT709Converter(\w+)\s*;")
CALL = re.compile(r"\b((?:BT709|Apple196|sRGB)_\w*convert\w+)\s*\(")
GAMMAFLAG = re.compile(r"\bapplyGammaMap\s*=\s*(\d)\s*;")


def parse_method(body):
    env = {}
    for m in ASSIGN.finditer(body):
        try:
            env[m.group(1)] = int(eval(m.group(2), {}, dict(env)))
        except Exception:
            pass
    expects = []
    for m in EXPECT.finditer(body):
        try:
            expects.append((m.group(1), int(eval(m.group(2), {}, dict(env)))))
        except Exception:
            expects.append((m.group(1), None))
    types = {m.group(1): m.group(2) for m in TYPE.finditer(body)}
    calls = []
    for m in CALL.finditer(body):
        if m.group(1) not in calls:
            calls.append(m.group(1))
    flag = GAMMAFLAG.search(body)
    return env, expects, types, calls, (int(flag.group(1)) if flag else None)


def triple(expects, names):
    d = {}
    for k, v in expects:
        if k in names and k not in d:
            d[k] = v
    if all(n in d and d[n] is not None for n in names):
        return [d[n] for n in names]
    return None


def collect_vectors():
    out = {"metal_decode": [], "converter": [], "direct_c": []}

    # 1. Metal decode tests: sRGB -> (Y,Cb,Cr) -> Metal decode, default Apple gamma
    f = os.path.join(TESTS, "MetalBT709DecoderTests.m")
    for name, line, body in methods(f):
        env, expects, types, _, _ = parse_method(body)
        ycc = triple(expects, ["Y", "Cb", "Cr"])
        rgb = triple(expects, ["Rout", "Gout", "Bout"])
        if ycc and rgb and all(k in env for k in ("Rin", "Gin", "Bin")):
            out["metal_decode"].append({
                "test": name, "src": "EmptyiOSTests/MetalBT709DecoderTests.m:%d" % line,
                "rgb_in": [env["Rin"], env["Gin"], env["Bin"]], "ycbcr": ycc, "rgb_out": rgb,
                "encode_type": types.get("type", types.get("encodeType")),
                "decode_type": types.get("decodeType")})

    # 2. converter tests (software / vImage / metal decode of greys)
    f = os.path.join(TESTS, "AppleEncodeDecodeBT709Tests.m")
    for name, line, body in methods(f):
        env, expects, types, _, _ = parse_method(body)
        ycc = triple(expects, ["Y", "Cb", "Cr"])
        rgb = triple(expects, ["Rout", "Gout", "Bout"])
        if ycc and rgb and all(k in env for k in ("Rin", "Gin", "Bin")):
            out["converter"].append({
                "test": name, "src": "EmptyiOSTests/AppleEncodeDecodeBT709Tests.m:%d" % line,
                "rgb_in": [env["Rin"], env["Gin"], env["Bin"]], "ycbcr": ycc, "rgb_out": rgb,
                "encode_type": types.get("encodeType", types.get("type")),
                "decode_type": types.get("decodeType")})

    # 3. direct calls into BT709.h
    f = os.path.join(TESTS, "CoreImageMetalFilterTests.m")
    for name, line, body in methods(f):
        env, expects, _, calls, flag = parse_method(body)
        ycc = triple(expects, ["Y", "Cb", "Cr"])
        rgb = triple(expects, ["R", "G", "B"])
        if ycc and all(k in env for k in ("Rin", "Gin", "Bin")) and len(calls) <= 2:
            out["direct_c"].append({
                "test": name, "src": "EmptyiOSTests/CoreImageMetalFilterTests.m:%d" % line,
                "rgb_in": [env["Rin"], env["Gin"], env["Bin"]], "ycbcr": ycc, "rgb_out": rgb,
                "calls": calls, "applyGammaMap": flag})
    # 4. 2x2 averaging tests: BT709_average_pixel_values(R1..B4, inputGamma, outputGamma) -> Y1..Y4, Cb, Cr
    #    (the encoder's per-block function, CoreImageMetalFilterTests.m:1683-2096)
    out["average_blocks"] = []
    gam = {"BT709GammaApple": 0, "BT709GammaSrgb": 1, "BT709GammaLinear": 2}
    for name, line, body in methods(f):
        if "BT709_average_pixel_values" not in body:
            continue
        vals = {m.group(1): int(m.group(2)) for m in re.finditer(r"\bint\s+([RGB][1-4])\s*=\s*(\d+)\s*;", body)}
        gi = re.search(r"inputGamma\s*=\s*(BT709Gamma\w+)", body)
        go = re.search(r"outputGamma\s*=\s*(BT709Gamma\w+)", body)
        _, expects, _, _, _ = parse_method(body)
        want = triple(expects, ["Y1", "Y2", "Y3", "Y4", "Cb", "Cr"])
        if len(vals) == 12 and gi and go and want:
            out["average_blocks"].append({
                "test": name, "src": "EmptyiOSTests/CoreImageMetalFilterTests.m:%d" % line,
                "rgb": [vals[c + str(i)] for i in range(1, 5) for c in "RGB"],
                "in": gam[gi.group(1)], "out": gam[go.group(1)], "y4cbcr": want})
    return out


def collect_histograms():
    """`XCTAssert([mDict[@"exact"] intValue] == N` plus the result comment block."""
    f = os.path.join(TESTS, "CoreImageMetalFilterTests.m")
    raw = open(f, encoding="utf-8", errors="replace").read()
    res = {}
    wanted = {"testConvertsRGBTo709_RoundTripAll_WithGamma": "itu709",
              "testConvertsRGBToApple196_RoundTripAll_WithGamma": "apple",
              "testConvertSRGBToSRGB_RoundTripAll_WithGamma": "srgb"}
    for name, key in wanted.items():
        pos = raw.index("(void)" + name)
        line = raw.count("\n", 0, pos) + 1
        tail = raw[pos:]
        exact = int(re.search(r'mDict\[@"exact"\] intValue\] == (\d+)', tail).group(1))
        block = re.search(r"/\*(.*?)\*/", tail, flags=re.S).group(1)
        hist = {}
        for m in re.finditer(r"(exact|off\d|offMore9)\s*=\s*(\d+)", block):
            hist[m.group(1)] = int(m.group(2))
        res[key] = {"src": "EmptyiOSTests/CoreImageMetalFilterTests.m:%d" % line,
                    "asserted_exact": exact, "comment_block": hist}
    return res


# ----------------------------------------------------------------- reference runs

def run_reference(ref):
    out = {"note": "produced by oracle/_ref/libbt709ref.so = /root/reference/Renderer/BT709.h + sRGB.h "
                   "compiled in place (gcc -O2 -ffp-contract=off, DEBUG undefined)"}
    out["table_sha256"] = {}
    out["histograms"] = {}
    out["thresholds_hex"] = {}
    names = {0: "apple", 1: "srgb", 2: "linear", 3: "itu709"}
    for g, n in names.items():
        t = ref.decode_table(g)
        out["table_sha256"][n] = hashlib.sha256(t.tobytes()).hexdigest()
        print("table", n, out["table_sha256"][n], flush=True)
        thr = ref.thresholds(g)
        out["thresholds_hex"][n] = ["%08x" % v for v in thr.view(np.uint32)]
        if g != 2:
            out["histograms"][n] = ref.roundtrip_histogram(g)
            print("hist", n, out["histograms"][n], flush=True)
    out["alpha_map"] = [ref.decode_alpha(a) for a in range(256)]
    # sampled subsample-block vectors (encode side; BT709_average_pixel_values)
    rng = np.random.default_rng(709)
    blocks = []
    for _ in range(64):
        rgb = [int(v) for v in rng.integers(0, 256, 12)]
        for ig, og in ((1, 0), (1, 1), (0, 0)):
            blocks.append({"rgb": rgb, "in": ig, "out": og, "y4cbcr": list(ref.subsample_block(rgb, ig, og))})
    out["subsample_blocks"] = blocks
    return out


# ----------------------------------------------------------------- bundled images

def load_bgra(path):
    from PIL import Image
    im = Image.open(path).convert("RGB")
    a = np.asarray(im, dtype=np.uint32)
    return (a[..., 0] << 16) | (a[..., 1] << 8) | a[..., 2]


def decode_ref(ref, gamma, y, uv):
    """Reference per-pixel decode over a tight NV12 crop via its full table."""
    tbl = decode_ref.tables.get(gamma)
    if tbl is None:
        tbl = decode_ref.tables[gamma] = ref.decode_table(gamma).reshape(-1, 3)
    h, w = y.shape
    cb = np.repeat(np.repeat(uv[:, 0::2], 2, axis=0), 2, axis=1).astype(np.uint32)
    cr = np.repeat(np.repeat(uv[:, 1::2], 2, axis=0), 2, axis=1).astype(np.uint32)
    idx = (y.astype(np.uint32) << 16) | (cb << 8) | cr
    rgb = tbl[idx.reshape(-1)].reshape(h, w, 3)
    out = np.empty((h, w, 4), dtype=np.uint8)
    out[..., 0] = rgb[..., 2]
    out[..., 1] = rgb[..., 1]
    out[..., 2] = rgb[..., 0]
    out[..., 3] = 0xFF
    return out


decode_ref.tables = {}


def collect_patterns(ref):
    imgs = [
        ("qt_hd", "Renderer/QuickTime_Test_Pattern_HD_sRGB.png",
         [(0, 0, 256, 128), (832, 476, 256, 128), (1664, 952, 256, 128), (448, 64, 128, 256)]),
        ("clouds", "Renderer/clouds_reflecting_off_the_beach-wallpaper-2048x1536.jpg",
         [(0, 0, 128, 128), (960, 704, 128, 128), (1920, 1408, 128, 128)]),
    ]
    arrays, meta = {}, []
    for key, rel, crops in imgs:
        bgra = load_bgra(os.path.join(REF, rel))
        H, W = bgra.shape
        for (x, y0, w, h) in crops:
            tile = np.ascontiguousarray(bgra[y0:y0 + h, x:x + w])
            # reference encoder: sRGB in -> Apple 1.96 video gamma out (the app's default clip type)
            yp, uv = ref.encode_nv12(tile.reshape(-1), w, h, 1, 0)
            tag = "%s_%d_%d_%dx%d" % (key, x, y0, w, h)
            arrays[tag + "_y"] = yp
            arrays[tag + "_uv"] = uv
            rec = {"tag": tag, "image": rel, "image_size": [W, H], "crop": [x, y0, w, h], "bgra_sha256": {}}
            for g, n in ((0, "apple"), (1, "srgb"), (2, "linear"), (3, "itu709")):
                out = decode_ref(ref, g, yp, uv)
                rec["bgra_sha256"][n] = hashlib.sha256(out.tobytes()).hexdigest()
                if g == 0:
                    arrays[tag + "_bgra_apple"] = out.reshape(h, w * 4)
            meta.append(rec)
            print("pattern", tag, flush=True)
    return arrays, meta


FULL_IMAGES = [
    ("qt_hd_srgb_full", "Renderer/QuickTime_Test_Pattern_HD_sRGB.png"),                # 1920 x 1080: BASELINE config 1's frame
    ("qt_hd_calibrated_full", "Renderer/QuickTime_Test_Pattern_HD_calibrated_RGB.png"),  # 1920 x 1080
    ("image_tga_full", "Renderer/Image.tga"),                                           # 512 x 512
    ("clouds_full", "Renderer/clouds_reflecting_off_the_beach-wallpaper-2048x1536.jpg"),  # 2048 x 1536 photograph (round 4)
]


def collect_full_patterns(ref):
    """Whole bundled images -> NV12 by the reference's encoder (sRGB in, Apple 1.96 out: the app's default) ->
    sha256 of the reference headers' decode in each gamma mode and of decode + exact 2:1 pass 2."""
    arrays, meta = {}, []
    for tag, rel in FULL_IMAGES:
        bgra = load_bgra(os.path.join(REF, rel))
        H, W = bgra.shape
        yp, uv = ref.encode_nv12(np.ascontiguousarray(bgra).reshape(-1), W, H, 1, 0)
        arrays[tag + "_y"], arrays[tag + "_uv"] = yp, uv
        rec = {"tag": tag, "image": rel, "image_size": [W, H], "nv12_sha256": hashlib.sha256(yp.tobytes() + uv.tobytes()).hexdigest(),
               "bgra_sha256": {}, "half_sha256": {}}
        for g, n in ((0, "apple"), (1, "srgb"), (2, "linear"), (3, "itu709")):
            rec["bgra_sha256"][n] = hashlib.sha256(decode_ref(ref, g, yp, uv).tobytes()).hexdigest()
            rec["half_sha256"][n] = hashlib.sha256(ref.decode_nv12_half(g, yp, uv).tobytes()).hexdigest()
        meta.append(rec)
        print("full pattern", tag, W, H, flush=True)
    return arrays, meta


def write_full_patterns(ref):
    arrays, meta = collect_full_patterns(ref)
    np.savez_compressed(os.path.join(HERE, "patterns_full.npz"), **arrays)
    json.dump({"note": "whole bundled images through the reference's encoder (cvpbu_ycbcr_subsample / "
                       "BT709_average_pixel_values, sRGB in, Apple out) and the reference headers' decode "
                       "(alpha byte 0xFF); half = decode + exact 2:1 pass 2 as in pass2.json",
               "images": meta}, open(os.path.join(HERE, "patterns_full.json"), "w"), indent=1)
    print("patterns_full.npz", os.path.getsize(os.path.join(HERE, "patterns_full.npz")), "bytes")


# ----------------------------------------------------------------- pass 2 (reference-composed)

def collect_pass2(ref, pattern_arrays):
    """sha256 (+ a few stored outputs) of decode + pass 2 computed by oracle/ref_harness.c's composition
    of the reference's own inlines: the pin for the oracle's and the GPU's fused rescale."""
    import pass2_cases as pc
    out = {"note": "ref_decode_nv12_half / ref_decode_nv12_scaled of oracle/ref_harness.c: pass 1 bytes by the "
                   "reference's per-pixel functions, sampler-side sRGB decode by sRGB_nonLinearNormToLinear(byteNorm(b)), "
                   "(((a+b)+c)+d)*0.25f or bilinear weights, sRGB_linearNormToNonLinear, (int)round(v*255.0f)",
           "cases": {}, "patterns": {}}
    stored = {}
    for kind, gamma, src, dst, seed, with_alpha in pc.CASES:
        y, c, a = pc.seeded_frame(src, seed, with_alpha)
        got = (ref.decode_nv12_half(gamma, y, c, alpha=a) if kind == "half"
               else ref.decode_nv12_scaled(gamma, y, c, dst[0], dst[1], alpha=a))
        key = pc.case_key(kind, gamma, src, dst, seed, with_alpha)
        out["cases"][key] = hashlib.sha256(got.tobytes()).hexdigest()
        if got.size <= 4096:
            stored[key] = got
    tags = sorted({k[:-2] for k in pattern_arrays if k.endswith("_y")})
    for tag in tags:
        y, c = pattern_arrays[tag + "_y"], pattern_arrays[tag + "_uv"]
        h, w = y.shape
        rec = {}
        for gamma in range(4):
            rec["half/g%d" % gamma] = hashlib.sha256(ref.decode_nv12_half(gamma, y, c).tobytes()).hexdigest()
            for (ow, oh) in pc.PATTERN_SIZES["scaled"]:
                rec["scaled/g%d/%dx%d" % (gamma, ow, oh)] = hashlib.sha256(
                    ref.decode_nv12_scaled(gamma, y, c, ow, oh).tobytes()).hexdigest()
        out["patterns"][tag] = rec
        print("pass2", tag, flush=True)
    # pass 1 into an RGBA16Float target: every (Y,Cb,Cr) -> three half codes, from the reference's matrix
    # step and curve functions (ref_rgba16f_table), index (Y<<16)+(Cb<<8)+Cr, little-endian uint16 R,G,B
    out["rgba16f_table_sha256"] = {}
    for g, n in ((0, "apple"), (1, "srgb"), (2, "linear"), (3, "itu709")):
        out["rgba16f_table_sha256"][n] = hashlib.sha256(ref.half_table(g).tobytes()).hexdigest()
        print("rgba16f table", n, out["rgba16f_table_sha256"][n], flush=True)
    out["alpha_half_map"] = [ref.alpha_half(a) for a in range(256)]
    return out, stored


def main():
    ref = Reference()
    if "--full-patterns" in sys.argv:
        write_full_patterns(ref)
        return
    vec = collect_vectors()
    vec["histograms"] = collect_histograms()
    print({k: len(v) for k, v in vec.items()})
    json.dump(vec, open(os.path.join(HERE, "vectors.json"), "w"), indent=1)

    r = run_reference(ref)
    arrays, meta = collect_patterns(ref)
    r["patterns"] = meta
    json.dump(r, open(os.path.join(HERE, "reference.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "patterns.npz"), **arrays)
    print("patterns.npz", os.path.getsize(os.path.join(HERE, "patterns.npz")), "bytes")

    p2, stored = collect_pass2(ref, arrays)
    json.dump(p2, open(os.path.join(HERE, "pass2.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "pass2_small.npz"), **{k.replace("/", "_"): v for k, v in stored.items()})
    print("pass2.json", len(p2["cases"]), "cases,", len(p2["patterns"]), "patterns")
    write_full_patterns(ref)


if __name__ == "__main__":
    main()
