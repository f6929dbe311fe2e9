"""rocprofv3 --kernel-trace --hip-trace --marker-trace CSVs of tools/gpu/owner_step_driver.py -> one ownership step as a
timeline: the kernels, and every HIP API call that WAITS for the device (synchronise calls, blocking copies) with the marker
interval it fell into.  The claim it checks: between step_begin and backward_exit nothing waits."""
import csv, glob, re, sys

d = sys.argv[1]
def load(pat):
    f = glob.glob(d + '/**/*' + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
kern, api, marks = load('kernel_trace.csv'), load('hip_api_trace.csv'), load('marker_api_trace.csv')
marks = [m for m in marks if m.get('Function') in ('step_begin', 'forward_exit', 'backward_exit', 'finish_exit', 'adam_exit')]
marks.sort(key=lambda m: int(m['Start_Timestamp']))
begins = [i for i, m in enumerate(marks) if m['Function'] == 'step_begin']
if len(begins) < 3:
    print('no marked steps found', len(marks)); sys.exit(0)
b = begins[-2]  # the last step but one: warm, and followed by another step_begin
step = marks[b:b + 6]
t0 = int(step[0]['Start_Timestamp'])
us = lambda t: (int(t) - t0) / 1000
print('# host-side marks of one step (us): ' + ', '.join(f"{m['Function']} {us(m['Start_Timestamp']):.0f}" for m in step))
waits = re.compile(r'Synchronize|hipMemcpy$|hipMemcpyDtoH$|hipMemcpyHtoD$|hipMemcpyDtoD$|hipMemset$|hipFree$|hipHostFree$|hipMalloc$|hipHostMalloc$')
t_end = int(step[-1]['Start_Timestamp'])
print('# HIP API calls of the step that can wait for the device:')
n_inside = 0
for a in sorted(api, key=lambda a: int(a['Start_Timestamp'])):
    s = int(a['Start_Timestamp'])
    if s < t0 or s > t_end or not waits.search(a['Function']):
        continue
    where = [m['Function'] for m in step if int(m['Start_Timestamp']) <= s][-1]
    dur = (int(a['End_Timestamp']) - s) / 1000
    inside = where in ('step_begin', 'forward_exit')
    n_inside += inside
    print(f"  {a['Function']:24s} at {us(s):8.1f} us, {dur:7.1f} us long, after `{where}`" + ("   <-- between forward entry and backward exit" if inside else ""))
print(f"# waiting calls between forward entry and backward exit: {n_inside}")
name = lambda r: re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", r['Kernel_Name'])).split("(")[0].split("<")[0].split("::")[-1][:34]
kern.sort(key=lambda r: int(r['Start_Timestamp']))
# the device runs behind the host: the step's kernels are those from its first cull pass to the next step's
culls = [i for i, r in enumerate(kern) if 'k_cull_compact' in r['Kernel_Name'] and int(r['Start_Timestamp']) >= t0]
if len(culls) >= 2:
    k0, k1 = culls[0], culls[1]
    g0 = int(kern[k0]['Start_Timestamp'])
    print(f"# the step on the device: {(int(kern[k1]['Start_Timestamp']) - g0) / 1000:.1f} us from cull to cull (rocprofv3 stretches kernels by ~12 %)")
    prev_end = g0
    for r in kern[k0:k1]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print(f"  {name(r):36s} start {(s - g0) / 1000:8.1f} dur {(e - s) / 1000:7.1f} gap {(s - prev_end) / 1000:6.1f} queue={r.get('Queue_Id')}")
        prev_end = max(prev_end, e)
