import os, sys, subprocess, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
if len(sys.argv) > 1:
    import numpy as np, helpers
    import beast_mcmc_amd as bm
    from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_NONE
    out = {}
    for (T, P, C) in [(3, 32, 1), (4, 32, 1), (5, 64, 1), (6, 128, 1), (8, 128, 1), (12, 128, 1), (12, 333, 1), (12, 333, 4), (12, 320, 4), (30, 1000, 4)]:
        wl = helpers.random_workload(T, P, 20, C, seed=7)
        t = BeagleTreeLikelihood(wl, rescaling=RESCALE_NONE, delay_rescaling=False)
        out["%d-%d-%d" % (T, P, C)] = t.getLogLikelihood()
        t.close()
    print(json.dumps(out))
else:
    a = json.loads(subprocess.run([sys.executable, __file__, "x"], capture_output=True, text=True).stdout.strip().splitlines()[-1])
    b = json.loads(subprocess.run([sys.executable, __file__, "x"], capture_output=True, text=True, env=dict(os.environ, BEAGLE_MI355_NO_T32_WALK="1")).stdout.strip().splitlines()[-1])
    for k in a: print(k, a[k], b[k], "OK" if abs(a[k]-b[k]) <= 1e-9*abs(b[k]) else "DIFF")
