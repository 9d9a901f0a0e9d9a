"""Where does the HOST stall in the first bench.py process on a fresh box?  (The GPU idles 18-32 ms once per iteration there and
not in a second process: profiles/r03_cold_first_process_gaps.txt.)

Runs bench.main() in-process with three probes:
  * faulthandler.dump_traceback_later(5 ms, repeat): a C thread that needs no GIL dumps every thread's Python stack; runs of
    identical main-thread stacks = a stall at that line;
  * gc.callbacks: every collection with generation and duration;
  * a watchdog thread that sleeps 1 ms and records jumps of its own clock (the whole process stopped, or the GIL was held).
python tools/cold_probe.py [bench.py flags]  ->  report on stdout after bench's JSON line."""
import faulthandler
import gc
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DUMP = "/tmp/cold_probe_stacks.txt"
PERIOD = 0.005


def main():
    t00 = time.perf_counter()
    gcs, jumps, run = [], [], [True]
    state = {}

    def gc_cb(phase, info):
        if phase == "start":
            state["t"] = time.perf_counter()
        else:
            gcs.append((time.perf_counter() - t00, info["generation"], time.perf_counter() - state.get("t", time.perf_counter()), info["collected"]))

    gc.callbacks.append(gc_cb)

    def watchdog():
        last = time.perf_counter()
        while run[0]:
            time.sleep(0.001)
            now = time.perf_counter()
            if now - last > 0.008:
                jumps.append((now - t00, now - last))
            last = now

    threading.Thread(target=watchdog, daemon=True).start()
    import bench          # (imports torch)

    f = open(DUMP, "w")
    t_arm = time.perf_counter() - t00
    faulthandler.dump_traceback_later(PERIOD, repeat=True, file=f, exit=False)
    sys.argv = ["bench.py"] + sys.argv[1:]
    try:
        bench.main()
    finally:
        faulthandler.cancel_dump_traceback_later()
        run[0] = False
        f.close()
    t_end = time.perf_counter() - t00
    # ---- report
    txt = open(DUMP).read()
    dumps = txt.split("Timeout (")[1:]
    print(f"probe: armed at {t_arm:.2f} s, ended at {t_end:.2f} s, {len(dumps)} stack dumps at {PERIOD * 1e3:.0f} ms "
          f"(expected {(t_end - t_arm) / PERIOD:.0f}: fewer = the dumper itself was stopped)")
    mains = []
    for d in dumps:
        blk = d.split("\n\n")
        m = [b for b in blk if "(most recent call first)" in b and ("Current thread" in b or "Thread 0x" in b)]
        main_blk = None
        for b in blk:
            if "bench.py" in b and "main" in b:
                main_blk = b
        lines = [ln.strip() for ln in (main_blk or "").splitlines() if ln.strip().startswith("File")]
        mains.append(tuple(lines[:6]))
    runs, i = [], 0
    while i < len(mains):
        j = i
        while j + 1 < len(mains) and mains[j + 1][:2] == mains[i][:2]:
            j += 1
        if j - i + 1 >= 3 and mains[i]:
            runs.append((j - i + 1, i, mains[i]))
        i = j + 1
    print(f"main-thread stalls (>= 3 consecutive identical top-2 frames = >= {3 * PERIOD * 1e3:.0f} ms), longest first:")
    for n, i, st in sorted(runs, reverse=True)[:15]:
        print(f"  {n * PERIOD * 1e3:6.0f} ms at dump {i} (t ~ {t_arm + i * PERIOD:.2f} s)")
        for ln in st[:5]:
            print("        " + ln[:170])
    print("gc collections (t, generation, seconds, collected):", [(round(t, 2), g, round(s, 4), c) for t, g, s, c in gcs if s > 0.002][:40])
    print("watchdog clock jumps > 8 ms (t, seconds):", [(round(t, 2), round(s, 3)) for t, s in jumps][:60])


if __name__ == "__main__":
    main()
