"""Timeline of lmm.pygemma's streamed SNP loop (pinned float32 X, n = 10 000, p = 100 000): run under
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -o t -- python3 tools/trace_loop.py run
then `python3 tools/trace_loop.py report <dir>` prints, for the LAST call's loop: every host->device copy of a batch (start, duration, GB/s),
the gaps between consecutive copies, and the kernel-busy time inside the loop window (where the loop's time goes)."""
import sys, glob, csv
import numpy as np
sys.path.insert(0, '/root/repo')
if sys.argv[1] == "run":
    from pygemma_amd import synth, lmm
    n, p, c = 10000, 100000, 5
    rng = np.random.default_rng(0)
    p_k = 4000
    GK = synth.genotypes(rng, n, p_k, np.float64)
    K = (GK @ GK.T / p_k).astype(np.float32)
    X = lmm.pinned_empty((n, p), np.float32)
    X[:] = synth.genotypes(rng, n, p)
    W = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, c - 1))], axis=1).astype(np.float32)
    y = (0.2 * X[:, 0] + GK @ (rng.standard_normal(p_k) * np.sqrt(0.5 / p_k)) + rng.standard_normal(n) * np.sqrt(0.5)).astype(np.float32).reshape(-1, 1)   # bench.py's phenotype: polygenic h2 = 0.5 + one causal SNP
    for rep in range(2):
        st = {}
        df = lmm.pygemma(y, X, W, K, stats=st)
        print(f"rep {rep}: loop {st['seconds']:.4f} s = {st['bytes_in'] / st['seconds'] / 1e9:.1f} GB/s; stages", {k: round(float(v), 4) for k, v in st.items() if k.endswith('_s')}, flush=True)
else:
    d = sys.argv[2]
    mc = list(csv.DictReader(open(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0])))
    kt = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
    big = [r for r in mc if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 2e6]      # the batch copies (pinned host memory shows up as DEVICE_TO_DEVICE)
    big.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = big[-12:]
    t0 = int(last[0]["Start_Timestamp"])
    prev_end = None
    tot = 0.0
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("H2D copy: start %8.3f ms  duration %6.3f ms%s" % ((s - t0) / 1e6, (e - s) / 1e6, "" if prev_end is None else "   gap since previous copy's end %6.3f ms" % ((s - prev_end) / 1e6)))
        prev_end = e; tot += (e - s) / 1e6
    t1 = max(int(r["End_Timestamp"]) for r in kt)
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in kt if int(r["Start_Timestamp"]) >= t0]
    busy = 0; cur_s = cur_e = None
    for s, e, _ in sorted(ks):
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("window from the first copy's start to the last kernel's end: %.2f ms; copies %.2f ms in total; kernels busy %.2f ms" % ((t1 - t0) / 1e6, tot, busy / 1e6))
    import collections
    byk = collections.defaultdict(float)
    for s, e, nme in ks: byk[nme[:50]] += (e - s) / 1e6
    for k_, v in sorted(byk.items(), key=lambda kv: -kv[1])[:6]: print("   %-52s %.2f ms" % (k_, v))
