"""Crash hunt for tests/test_gpu_fuzz.py::test_many_processes_share_the_device (round 5's driver run lost child 4,
case 704000069, to a SIGSEGV).  Runs the same 12 processes x 45 s as the test, round after round, every child under
`python -X faulthandler` with profiles/tools/segv_trace.c preloaded and `ulimit -c unlimited`; a child that dies
leaves (a) python's stacks of every thread, (b) the native backtrace of the faulting thread and the memory map,
(c) a core that rocgdb turns into `thread apply all bt`.  Everything lands in <out>/.

    python profiles/tools/segv_hunt.py <out dir> <minutes> [procs=12] [seconds=45] [extra env K=V ...]"""
import glob
import os
import resource
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.abspath(sys.argv[1])
minutes = float(sys.argv[2])
n_procs = int(sys.argv[3]) if len(sys.argv) > 3 else 12
secs = sys.argv[4] if len(sys.argv) > 4 else "45"
extra = dict(a.split("=", 1) for a in sys.argv[5:])
os.makedirs(out, exist_ok=True)
so = "/tmp/segv_trace.so"
subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "profiles", "tools", "segv_trace.c")])
resource.setrlimit(resource.RLIMIT_CORE, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
t_end = time.time() + 60 * minutes
rnd = deaths = runs = 0
log = open(os.path.join(out, "hunt.log"), "a")
def say(*a):
    print(*a, file=log, flush=True); print(*a, flush=True)
say("core_pattern:", open("/proc/sys/kernel/core_pattern").read().strip(), "extra env:", extra)
while time.time() < t_end and deaths < 4:
    # round 0 is the driver's own seeds; later rounds move on
    base = 700_000_000 + 20_000_000 * rnd
    procs = []
    for p in range(n_procs):
        wd = f"/tmp/hunt/r{rnd}p{p}"
        shutil.rmtree(wd, ignore_errors=True); os.makedirs(wd)
        env = dict(os.environ, FUZZ_SHARDS="1", FUZZ_SEED_SHARDS="1", FUZZ_TRACE=os.path.join(wd, "case"),
                   PYTHONPATH=ROOT + ":" + os.path.join(ROOT, "tests"), LD_PRELOAD=so,
                   SEGV_TRACE_FILE=os.path.join(wd, "native.txt"), FUZZ_ROOT=ROOT, **extra)
        procs.append((wd, subprocess.Popen([sys.executable, "-X", "faulthandler", os.path.join(ROOT, "tests", "fuzz_open_ended.py"),
                                            secs, str(base + 1_000_000 * p)], cwd=ROOT, env=env,
                                           stdout=open(os.path.join(wd, "out"), "w"), stderr=open(os.path.join(wd, "err"), "w"))))
    for p, (wd, pr) in enumerate(procs):
        rc = pr.wait()
        runs += 1
        o = open(os.path.join(wd, "out")).read()
        if rc == 0 and "fuzz ok" in o:
            continue
        deaths += 1
        tag = f"death{deaths}_r{rnd}p{p}"
        case = open(os.path.join(wd, "case")).read() if os.path.exists(os.path.join(wd, "case")) else "?"
        say(f"DEATH round {rnd} proc {p} rc {rc} case {case}")
        with open(os.path.join(out, tag + ".txt"), "w") as f:
            f.write(f"rc {rc} case {case}\n--- stdout\n{o[-3000:]}\n--- stderr (faulthandler)\n")
            f.write(open(os.path.join(wd, "err")).read()[-20000:])
            if os.path.exists(os.path.join(wd, "native.txt")):
                f.write("\n--- native backtrace of the faulting thread + maps\n" + open(os.path.join(wd, "native.txt")).read()[-60000:])
        cores = glob.glob(os.path.join(ROOT, "core*")) + glob.glob(os.path.join(wd, "core*")) + glob.glob("/tmp/core*")
        say("cores:", cores)
        for c in cores[:1]:
            try:
                g = subprocess.run(["/opt/rocm/bin/rocgdb", "-batch", "-ex", "set pagination off", "-ex", "info sharedlibrary",
                                    "-ex", "thread apply all bt 40", "-ex", "info registers", sys.executable, c],
                                   capture_output=True, text=True, timeout=600)
                open(os.path.join(out, tag + "_gdb.txt"), "w").write(g.stdout[-400000:] + "\n--- stderr\n" + g.stderr[-5000:])
            except Exception as e:                      # noqa: BLE001
                say("rocgdb failed:", e)
        for c in cores:
            os.remove(c)
    say(f"round {rnd} done: {runs} process-runs, {deaths} deaths, {time.time() - (t_end - 60 * minutes):.0f} s")
    rnd += 1
say(f"hunt over: {runs} process-runs of {secs} s, {deaths} deaths")
