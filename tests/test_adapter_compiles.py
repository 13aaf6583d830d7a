"""adapters/viso_hip_adapter.inc through a C++ front end.

The adapter is compiled by a libviso maintainer INSIDE the reference's src/viso.cpp, against OpenCV
and Boost — neither exists in this image, so until now no compiler had ever seen it.  Here the translation unit is
rebuilt the way the patched viso.cpp presents it to the adapter: the reference's own declarations in front of the
include — typedefs + struct param (patched src/viso.h), struct MatchParams, kp2mat, the two triangulate_rectified
templates (patched src/viso.cpp) — are CUT FROM THE PATCHED REFERENCE AT TEST TIME (never committed), the OpenCV / Boost
names they and the adapter use come from tests/adapter_stub/cv_boost_stub.hpp (a type stub, nothing is executed), then
`#include "viso_hip_adapter.inc"` exactly as the patch places it, then callers with the reference's call-site argument
types (src/viso.cpp:1240,1264,1275,1282,1245-1247,1313,1316).  `g++ -std=c++11 -fsyntax-only -Wall -Werror`.
This is a type check of the adapter, not an oracle.  Skipped where /root/reference does not exist (the GPU box)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PATCH = os.path.join(ROOT, "adapters", "libviso_hip.patch")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present")


def _block(lines, start_rx, end_rx, start_from=0):
    a = next(i for i in range(start_from, len(lines)) if re.match(start_rx, lines[i]))
    z = next(i for i in range(a, len(lines)) if re.match(end_rx, lines[i]))
    return a, z


CALLERS = r'''
// the reference's call sites, argument types as in sequence_odometry (src/viso.cpp:1240-1316)
static void use_everything(const Mat& P_F) {
    KeyPoints kp1, kp2, kp1_prev;  Mat d1, d2, d1_prev;  Matches match_lr, match_lr_prev, match11, match22, match_pcl;
    vector<Vec4i> circ_match;  struct param param;  Mat x, X, X_prev, observe, tr_mat;
    MatchParams sp(P_F);
    match_desc(kp1, kp2, d1, d2, match_lr, sp);                                   // :1240
    match_desc(kp1, kp1_prev, d1, d1_prev, match11);                              // :1264 (default argument)
    match_circle(match_lr, match_lr_prev, match11, match22, circ_match, match_pcl);   // :1282
    collect_matches(kp1, kp2, match_lr, x);                                       // :1245
    X = triangulate_rectified<double>(x, param);                                  // :1247
    vector<double> tr(6, 0);  vector<int> inliers;
    bool ok = ransac_minimize_reproj(X_prev, observe, tr, inliers, param);        // :1313
    ok = minimize_reproj(X_prev, observe, tr, param, inliers) && ok;
    tr2mat(tr, tr_mat);                                                           // :1316
    param.frame_index = 7; param.ransac_seed = 1;
    (void)ok;
}
'''


def test_adapter_type_checks_in_the_patched_reference(tmp_path):
    dst = str(tmp_path / "libviso")
    shutil.copytree(REF, dst)
    r = subprocess.run(["git", "apply", PATCH], cwd=dst, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    h = open(os.path.join(dst, "src", "viso.h"), encoding="utf-8", errors="surrogateescape").read().splitlines()
    c = open(os.path.join(dst, "src", "viso.cpp"), encoding="utf-8", errors="surrogateescape").read().splitlines()
    a, _ = _block(h, r"^typedef vector<KeyPoint> KeyPoints;", r"^typedef pair<Mat,Mat> image_pair;")
    _, z = _block(h, r"^struct param\s*$", r"^};")
    decl_h = h[a:z + 1]
    assert any("frame_index" in ln for ln in decl_h)                       # the patched struct param
    a, z = _block(c, r"^struct MatchParams\s*$", r"^};")
    match_params = c[a:z + 1]
    a, _ = _block(c, r"^kp2mat\(const KeyPoints& kp\)", r"^kp2mat")
    _, z = _block(c, r"^kp2mat\(const KeyPoints& kp\)", r"^}")
    kp2mat = c[a - 1:z + 1]                                                # with its return type line
    a, _ = _block(c, r"^triangulate_rectified\(const Mat& x, /\* coords", r".*")
    inc = next(i for i, ln in enumerate(c) if ln.strip() == '#include "viso_hip_adapter.inc"')
    tri_and_adapter = c[a - 1:inc + 2]                                     # both templates, the #ifdef, the include, its #endif
    assert tri_and_adapter[0].startswith("template<typename T> Mat") and tri_and_adapter[-1].startswith("#endif")
    tu = "\n".join(
        ['#include "cv_boost_stub.hpp"', "using namespace std;",
         "using cv::Mat; using cv::KeyPoint; using cv::Point2f; using cv::Vec3i; using cv::Vec4i; using cv::DataType;"]
        + decl_h + match_params + kp2mat + tri_and_adapter + [CALLERS])
    src = tmp_path / "adapter_tu.cpp"
    src.write_text(tu, encoding="utf-8", errors="surrogateescape")
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Werror", "-Wno-sign-compare", "-Wno-unused-function", "-Wno-reorder", "-DVISO_USE_HIP",
           "-I", os.path.join(ROOT, "tests", "adapter_stub"), "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "adapters"), str(src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-6000:]
    # the check has teeth: a wrong argument type inside the adapter is an error in this translation unit
    bad_dir = tmp_path / "broken"
    bad_dir.mkdir()
    a_txt = open(os.path.join(ROOT, "adapters", "viso_hip_adapter.inc")).read()
    broken = a_txt.replace("viso_tr2mat(tr.data(), Tr.ptr<double>());", "viso_tr2mat(tr, Tr.ptr<double>());")
    assert broken != a_txt
    (bad_dir / "viso_hip_adapter.inc").write_text(broken)
    cmd2 = [c for c in cmd if c != os.path.join(ROOT, "adapters")]
    cmd2 = cmd2[:-1] + [str(bad_dir), str(src)]   # the "-I" that pointed at adapters/ now takes the broken copy's directory
    r = subprocess.run(cmd2, capture_output=True, text=True)
    assert r.returncode != 0 and "viso_tr2mat" in r.stderr
