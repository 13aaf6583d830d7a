"""A synthetic KITTI odometry tree for the driver tests: $KITTI_HOME/sequences/<seq>/{calib.txt, image_0/%06d.png,
image_1/%06d.png} (reference src/kitti.cpp:96-110) from libviso_amd.synth images."""
import os

import pngutil


def write_tree(home, seq_name, seq, first_index=0):
    """seq: synth.make_image_sequence(...) -> returns the sequence directory."""
    base = os.path.join(home, "sequences", seq_name)
    nf = seq["images"].shape[0]
    for side in (0, 1):
        os.makedirs(os.path.join(base, f"image_{side}"), exist_ok=True)
        for t in range(nf):
            # KITTI's own layout: image_0/%06d.png, 8-bit grayscale (src/kitti.cpp:108-110)
            pngutil.write_gray_png(os.path.join(base, f"image_{side}", "%06d.png" % (first_index + t)), seq["images"][t, side])
    with open(os.path.join(base, "calib.txt"), "w") as f:      # src/kitti.cpp:23-46: reads the lines P0: and P1:
        for name, P in (("P0", seq["P1"]), ("P1", seq["P2"]), ("P2", seq["P1"]), ("P3", seq["P2"])):
            f.write(name + ": " + " ".join("%.12e" % v for v in P.reshape(-1)) + "\n")
    return base
