import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    bad = a[k] != b[k]
    if bad.any(): print(k, 'differs in', int(bad.sum()), 'first', np.argwhere(bad)[0].tolist())
print('compared', len(a.files))
