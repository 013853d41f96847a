import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from mad_icp_amd import capi, synth
scene = synth.Scene(0)
s = synth.render_scan(scene, synth.path_pose(0.0), 100)
ctx = capi.Context(0)
for rep in range(3):
    tb=[];tu=[];tr=[];tt=[]
    for i in range(12):
        t=time.perf_counter(); ht = capi.HostTree(s, 0.2, 0.1, 4); tb.append(time.perf_counter()-t)
        nodes = ht.nodes
        t=time.perf_counter(); tid = ctx.tree_upload(nodes, ht.num_leaves); tu.append(time.perf_counter()-t)
        t=time.perf_counter(); ctx.tree_transform(tid, np.eye(3), np.zeros(3)); tt.append(time.perf_counter()-t)
        t=time.perf_counter(); ctx.tree_release(tid); tr.append(time.perf_counter()-t)
    print("build %.2f ms  upload %.3f ms  transform %.3f ms  release %.3f ms  (nodes %d)" % (1e3*np.median(tb), 1e3*np.median(tu), 1e3*np.median(tt), 1e3*np.median(tr), ht.num_nodes))
