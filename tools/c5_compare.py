import numpy as np
a = np.load("/root/repo/gpurun_out/c5_new.npy"); b = np.load("/root/repo/gpurun_out/c5_old.npy")
print("C5 new vs old kernels: max|dE|/lam_max %.2e   worst relative %.2e at E=%.3e" % (
    np.max(np.abs(a - b)) / np.max(np.abs(b)), np.max(np.abs(a - b) / np.abs(b)), b[np.argmax(np.abs(a - b) / np.abs(b))]))
