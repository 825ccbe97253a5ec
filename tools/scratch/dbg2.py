import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import helpers
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_NONE
T, P, C = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
wl = helpers.random_workload(T, P, 20, C, seed=7)
print("left", wl.tree.left, "right", wl.tree.right)
t = BeagleTreeLikelihood(wl, rescaling=RESCALE_NONE, delay_rescaling=False)
print(t.getLogLikelihood())
