#!/usr/bin/env python3
"""Generates adapters/libviso_hip.patch against a checkout of alexkreimer/libviso (default /root/reference).

The patch (a) guards the reference's own definitions of the hot-path functions with `#ifndef VISO_USE_HIP`, (b)
#includes adapters/viso_hip_adapter.inc inside src/viso.cpp right behind the triangulate_rectified templates, (c) adds
the RANSAC stream key (ransac_seed, frame_index) to struct param and sets frame_index per frame in
sequence_odometry, (d) adds the VISO_USE_HIP option to src/CMakeLists.txt.  Every edit is anchored on the function's
first line (checked against the expected text) and ends at that function's closing brace, found by brace counting.
Reads the reference as text only; writes nothing there.

    python tools/make_adapter_patch.py [reference_root] > adapters/libviso_hip.patch
"""
import difflib
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


def function_end(lines, start):
    """Index of the line holding the closing brace of the function whose signature starts at `start`."""
    depth, seen = 0, False
    for i in range(start, len(lines)):
        for ch in re.sub(r"//.*", "", lines[i]):
            if ch == "{":
                depth += 1
                seen = True
            elif ch == "}":
                depth -= 1
                if seen and depth == 0:
                    return i
    raise SystemExit(f"no closing brace after line {start + 1}")


def guard(lines, anchor_regex, back=1, occurrence=0):
    """Wrap the function whose name line matches anchor_regex (the return type sits `back` lines above)."""
    hits = [i for i, ln in enumerate(lines) if re.match(anchor_regex, ln)]
    if len(hits) <= occurrence:
        raise SystemExit(f"anchor not found: {anchor_regex}")
    a = hits[occurrence] - back
    z = function_end(lines, hits[occurrence])
    lines[z] = lines[z] + "#endif /* !VISO_USE_HIP */\n"
    lines[a] = "#ifndef VISO_USE_HIP /* libviso_hip.so takes this one: viso_hip_adapter.inc */\n" + lines[a]
    return z


def patched_viso_cpp(text):
    L = text.splitlines(keepends=True)
    guard(L, r"^tr2mat\(vector<double> tr,Mat& Tr\)")                                   # :109-133
    guard(L, r"^match_circle\(const Matches& match_lr, const Matches& match_lr_prev,")    # :207-243
    # collect_matches: three overloads; the one that fills `Mat &x` (:501-514) is the third
    cm = [i for i, ln in enumerate(L) if re.match(r"^collect_matches\(const KeyPoints& kp1, const KeyPoints &kp2,", ln)]
    third = [i for i in cm if "Mat &x)" in L[i + 1]]
    if len(third) != 1:
        raise SystemExit("collect_matches(.., Mat &x) not found")
    z = function_end(L, third[0])
    L[z] += "#endif /* !VISO_USE_HIP */\n"
    L[third[0] - 1] = "#ifndef VISO_USE_HIP /* libviso_hip.so takes this one: viso_hip_adapter.inc */\n" + L[third[0] - 1]
    guard(L, r"^match_desc\(const KeyPoints& kp1, const KeyPoints& kp2,", back=2)          # :668-726 (with its comment line)
    # the adapter goes behind the second triangulate_rectified template (:1156-1162)
    tpl = [i for i, ln in enumerate(L) if re.match(r"^triangulate_rectified\(const Mat& x,\s*$", ln)]
    if len(tpl) != 1:
        raise SystemExit("triangulate_rectified(x, param) template not found")
    z = function_end(L, tpl[0])
    L[z] += ("#ifdef VISO_USE_HIP /* match_desc, match_circle, collect_matches, triangulate_rectified<double>, minimize_reproj,\n"
             "                       ransac_minimize_reproj, tr2mat on libviso_hip.so (MI355X) */\n"
             "#include \"viso_hip_adapter.inc\"\n#endif\n")
    # frame index = the RANSAC stream key of this frame
    it = [i for i, ln in enumerate(L) if 'BOOST_LOG_TRIVIAL(info) << "iter: " << iter_num;' in ln]
    so = [i for i, ln in enumerate(L) if re.match(r"^sequence_odometry\(const Mat& P1, const Mat& P2, StereoImageGenerator& images,", ln)]
    if len(so) != 1:
        raise SystemExit("sequence_odometry not found")
    it = [i for i in it if so[0] < i < function_end(L, so[0])]                            # :1207 (not calibratedSFM's :1350)
    if len(it) != 1:
        raise SystemExit("sequence_odometry loop head not found")
    L[it[0]] += ("#ifdef VISO_USE_HIP\n        param.frame_index = images.index() - 1; /* RANSAC stream key of this frame: its file number (begin + iter_num) */\n"
                 "#endif\n")
    guard(L, r"^ransac_minimize_reproj\(const Mat& X, /\* 3d points \*/")                 # :1543-1580
    guard(L, r"^minimize_reproj\(const Mat& X, const Mat& observe, vector<double>& tr,")  # :1583-1623
    return "".join(L)


def patched_viso_h(text):
    L = text.splitlines(keepends=True)
    i = [k for k, ln in enumerate(L) if re.match(r"^\s*bool save_debug;", ln)]
    if len(i) != 1:
        raise SystemExit("struct param not found")
    L[i[0]] += ("#ifdef VISO_USE_HIP\n    /* deterministic replacement for randomsample's random_device: stream key of the RANSAC triples */\n"
                "    unsigned long long ransac_seed = 0, frame_index = 0;\n#endif\n")
    # the generator's file index, so that the stream key can be the ABSOLUTE frame number (what viso_kitti keys on)
    c = [k for k, ln in enumerate(L) if re.match(r"^class StereoImageGenerator\s*$", ln)]
    if len(c) != 1:
        raise SystemExit("class StereoImageGenerator not found")
    pr = [k for k in range(c[0], len(L)) if re.match(r"^private:", L[k])]
    L[pr[0]] = ("#ifdef VISO_USE_HIP\n    int index() const { return m_index; } /* file number of the NEXT frame */\n#endif\n" + L[pr[0]])
    return "".join(L)


def patched_cmake(text):
    L = text.splitlines(keepends=True)
    i = [k for k, ln in enumerate(L) if ln.startswith("add_library(viso ")]
    if len(i) != 1:
        raise SystemExit("add_library(viso ...) not found")
    L[i[0]] = ("option(VISO_USE_HIP \"hot path on libviso_hip.so (MI355X)\" OFF)\n"
               "set(VISO_HIP_DIR \"\" CACHE PATH \"checkout of libviso_amd (include/, adapters/, libviso_amd/libviso_hip.so)\")\n"
               + L[i[0]] +
               "if(VISO_USE_HIP)\n"
               "  target_compile_definitions(viso PUBLIC VISO_USE_HIP)\n"
               "  target_include_directories(viso PUBLIC ${VISO_HIP_DIR}/include ${VISO_HIP_DIR}/adapters)\n"
               "  target_link_libraries(viso ${VISO_HIP_DIR}/libviso_amd/libviso_hip.so)\n"
               "endif()\n")
    return "".join(L)


def main():
    out = []
    for rel, fn in (("src/CMakeLists.txt", patched_cmake), ("src/viso.cpp", patched_viso_cpp), ("src/viso.h", patched_viso_h)):
        old = open(os.path.join(REF, rel), encoding="utf-8", errors="surrogateescape").read()
        new = fn(old)
        out += difflib.unified_diff(old.splitlines(keepends=True), new.splitlines(keepends=True),
                                    "a/" + rel, "b/" + rel, n=1)
    sys.stdout.write("".join(out))


if __name__ == "__main__":
    main()
