import csv, statistics, collections, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
lin=sorted([r for r in rows if "linearize" in r["Kernel_Name"]], key=lambda r:int(r["Start_Timestamp"]))
sol=sorted([r for r in rows if "icp_solve" in r["Kernel_Name"]], key=lambda r:int(r["Start_Timestamp"]))
n=int(sys.argv[2])*15
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in lin[:n]]
e=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in sol[:n]]
print("timed-region launches: linearize avg %.2f us, solve avg %.2f us (n=%d)"%(statistics.mean(d), statistics.mean(e), n))
