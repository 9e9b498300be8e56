cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c4p -o c4 -- python $GRAFT_REPO_ROOT/tools/edsr_b16.py 128 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/c4p/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("sum ms/step", sum(float(r["TotalDurationNs"]) for r in rows)/5e6)
for r in rows[:16]:
    print("%5s x %9.1f us (min %8.1f max %8.1f) %5.1f%%  %s"%(r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, float(r["Percentage"]), r["Name"][:90]))
PY
python - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/c4p/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last step: print kernels > 100 us in order
n=len(rows)//5
for r in rows[-n:]:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    if d>60: print("%8.1f us  %s grid %s"%(d, r["Kernel_Name"][:80], r.get("Grid_Size_X", r.get("Grid_Size",""))))
PY
