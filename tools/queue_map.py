import csv, sys, collections
f = sys.argv[1]
q = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    k = "icp_publish" if "icp_publish" in n else "icp_*" if "icp_" in n else "tb_*" if "tb_" in n else "moving_from_leaves" if "moving_from" in n else "other"
    q[k][r.get("Queue_Id", "?")] += 1
for k, c in q.items():
    print(k, dict(c))
