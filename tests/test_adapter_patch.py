"""adapters/libviso_hip.patch must apply to the reference tree, leave it untouched when VISO_USE_HIP is not defined, and
with the macro swap exactly the hot-path functions for adapters/viso_hip_adapter.inc.  The reference
itself cannot be compiled in this image (no OpenCV / Boost), so this is a source-level check: `git apply --check` on a
scratch copy, then the two preprocessor views of the patched files.  Skipped where /root/reference does not exist (the
GPU box)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PATCH = os.path.join(ROOT, "adapters", "libviso_hip.patch")
ADAPTER = os.path.join(ROOT, "adapters", "viso_hip_adapter.inc")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present")

REPLACED = {   # function -> first line of its definition in the reference's viso.cpp
    "tr2mat": r"^tr2mat\(vector<double> tr,Mat& Tr\)",
    "match_circle": r"^match_circle\(const Matches& match_lr, const Matches& match_lr_prev,",
    "match_desc": r"^match_desc\(const KeyPoints& kp1, const KeyPoints& kp2,",
    "ransac_minimize_reproj": r"^ransac_minimize_reproj\(const Mat& X, /\* 3d points \*/",
    "minimize_reproj": r"^minimize_reproj\(const Mat& X, const Mat& observe, vector<double>& tr,",
}


def view(text, defined):
    """The file as the preprocessor passes it on, for the VISO_USE_HIP conditionals only (every other directive is kept
    as text; nesting is tracked so that an #endif of another conditional inside a guarded function is not mistaken)."""
    out, stack = [], []          # stack entries: None = foreign conditional, True/False = ours (keep / drop)
    for ln in text.splitlines(keepends=True):
        s = ln.strip()
        ours = re.match(r"#\s*(ifdef|ifndef)\s+VISO_USE_HIP\b", s)
        if ours:
            stack.append((ours.group(1) == "ifdef") == defined)
            continue
        if re.match(r"#\s*if", s):
            stack.append(None)
        elif re.match(r"#\s*endif", s):
            top = stack.pop()
            if top is not None:
                continue
        if all(t is not False for t in stack):
            out.append(ln)
    assert not stack
    return "".join(out)


@pytest.fixture(scope="module")
def patched(tmp_path_factory):
    dst = str(tmp_path_factory.mktemp("ref") / "libviso")
    shutil.copytree(REF, dst)
    r = subprocess.run(["git", "apply", "--check", PATCH], cwd=dst, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(["git", "apply", PATCH], cwd=dst, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return dst


def _read(root, rel):
    return open(os.path.join(root, rel), encoding="utf-8", errors="surrogateescape").read()


def test_patch_is_what_the_generator_writes():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_adapter_patch.py"), REF], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == open(PATCH).read()


def test_without_the_macro_the_tree_is_the_reference(patched):
    for rel in ("src/viso.cpp", "src/viso.h"):
        assert view(_read(patched, rel), defined=False) == _read(REF, rel), rel
    # every other source file is untouched
    for rel in ("src/kitti.cpp", "src/mvg.cpp", "src/mvg.h", "src/estimation.cpp", "src/estimation.h", "src/misc.h", "test/test.cpp"):
        assert _read(patched, rel) == _read(REF, rel), rel


def test_with_the_macro_the_hot_path_comes_from_the_adapter(patched):
    cpp = view(_read(patched, "src/viso.cpp"), defined=True)
    lines = cpp.splitlines()
    for name, rx in REPLACED.items():
        assert not [ln for ln in lines if re.match(rx, ln)], f"the reference's {name} is still compiled"
    # collect_matches: the (.., Mat &x) overload is gone, the other two stay
    cm = [i for i, ln in enumerate(lines) if ln.startswith("collect_matches(const KeyPoints& kp1, const KeyPoints &kp2,")]
    assert len(cm) == 2 and not any("Mat &x)" in lines[i + 1] for i in cm)
    # the adapter: once, behind both triangulate_rectified templates (an explicit specialisation needs its primary
    # template), in front of sequence_odometry (its first use), where MatchParams and kp2mat are visible
    inc = [i for i, ln in enumerate(lines) if ln.strip() == '#include "viso_hip_adapter.inc"']
    assert len(inc) == 1
    pos = lambda rx: [i for i, ln in enumerate(lines) if re.match(rx, ln)]
    assert max(pos(r"^triangulate_rectified\(const Mat& x,")) < inc[0] < pos(r"^sequence_odometry\(const Mat& P1")[0]
    assert pos(r"^struct MatchParams")[0] < inc[0] and pos(r"^kp2mat\(const KeyPoints& kp\)")[0] < inc[0]
    # the call sites are the reference's own, unchanged (5-argument match_desc needs the adapter's default argument)
    assert sum("match_desc(kp1,kp1_prev,d1,d1_prev,match11);" in ln for ln in lines) == 1
    # RANSAC stream key = the frame's FILE NUMBER (what viso_kitti / kitti_shard key on), read off the generator
    assert sum("param.frame_index = images.index() - 1;" in ln for ln in lines) == 1
    h = view(_read(patched, "src/viso.h"), defined=True)
    assert "unsigned long long ransac_seed = 0, frame_index = 0;" in h
    assert "int index() const { return m_index; }" in h
    assert "VISO_USE_HIP" in _read(patched, "src/CMakeLists.txt")


def test_adapter_defines_each_replacement_once():
    a = open(ADAPTER).read()
    for sig in (r"^void match_desc\(const KeyPoints& kp1, const KeyPoints& kp2, const Descriptors& d1, const Descriptors& d2,\n\s+Matches& match, const MatchParams& sp = MatchParams\(\)\)",
                r"^bool minimize_reproj\(const Mat& X, const Mat& observe, vector<double>& tr, const struct param& param,",
                r"^bool ransac_minimize_reproj\(const Mat& X, const Mat& observe, vector<double>& best_tr,",
                r"^void match_circle\(const Matches& lr, const Matches& lrp, const Matches& m11, const Matches& m22,",
                r"^void tr2mat\(vector<double> tr, Mat& Tr\)",
                r"^void collect_matches\(const KeyPoints& kp1, const KeyPoints& kp2, const Matches& match, Mat& x\)",
                r"^template <> Mat triangulate_rectified<double>\(const Mat& x, const struct param& param\)"):
        assert len(re.findall(sig, a, flags=re.M)) == 1, sig
    assert "static uint64_t call" not in a and "param.frame_index" in a and '#include "viso.h"' not in a
