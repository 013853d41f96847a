"""Device front-end vs host path over a long synthetic drive: trajectory error against GROUND TRUTH for both, the
deviation between them, keyframe decisions.  python tools/frontend_drive.py [frames] [deskew 0/1] [step_m]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mad_icp_amd import _build, synth  # noqa: E402

_build.build_pybind()
from mad_icp.src.pybind import pypeline as m  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
deskew = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
step = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
scene = synth.Scene(0)
args = (10.0, deskew, 0.2, 0.1, 0.8, 0.1, 0.02, 16, 16, False)
host, dev = m.Pipeline(*args), m.Pipeline(*args)
dev.setDeviceFrontEnd(True)
T0inv = np.linalg.inv(synth.path_pose(0.0))
eh, ed, dev_host, kf_mismatch, id_h, id_d = [], [], [], 0, [], []
t0 = time.time()
for i in range(n):
    sc = synth.render_scan(scene, synth.path_pose(step * i), 100 + i)
    host.compute(0.1 * i, sc)
    dev.compute(0.1 * i, sc)
    gt = T0inv @ synth.path_pose(step * i)
    Th, Td = np.asarray(host.currentPose()), np.asarray(dev.currentPose())
    eh.append(np.linalg.norm((np.linalg.inv(gt) @ Th)[:3, 3]))
    ed.append(np.linalg.norm((np.linalg.inv(gt) @ Td)[:3, 3]))
    dev_host.append(np.linalg.norm((np.linalg.inv(Th) @ Td)[:3, 3]))
    kf_mismatch += int(host.keyframeID() != dev.keyframeID())
    id_h.append(host.keyframeID())
    id_d.append(dev.keyframeID())
eh, ed, dev_host = np.array(eh), np.array(ed), np.array(dev_host)
print("frames %d deskew %s step %.2f m (%.1f s)" % (n, deskew, step, time.time() - t0))
print("host path   : final error %.4f m  rms %.4f m  max %.4f m" % (eh[-1], np.sqrt((eh ** 2).mean()), eh.max()))
print("device front: final error %.4f m  rms %.4f m  max %.4f m" % (ed[-1], np.sqrt((ed ** 2).mean()), ed.max()))
print("device vs host: max %.4f m  rms %.4f m ; frames with different keyframe id: %d ; keyframes promoted host %d dev %d" % (
    dev_host.max(), np.sqrt((dev_host ** 2).mean()), kf_mismatch, len(set(id_h)), len(set(id_d))))
