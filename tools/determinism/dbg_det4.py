import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = a["rew0"] != b["rew0"]
print("rew0 differing (step,row):", np.argwhere(bad)[:12].tolist())
t, row = np.argwhere(bad)[0]; e = row // 2
for nm, x in (("A", a), ("B", b)):
    print(nm, "rew", x["rew0"][t - 1:t + 2, 2 * e:2 * e + 2].tolist(), "done", x["done0"][t - 1:t + 2, 2 * e].tolist(), "act", x["act0"][t - 1:t + 2, 2 * e:2 * e + 2].tolist())
for k in (t, t + 1):
    d = np.argwhere(a["obs0"][k, 2 * e] != b["obs0"][k, 2 * e]).ravel()
    print("obs step", k, "row", 2 * e, "differing features", d.tolist()[:40])
    print("  A", np.round(a["obs0"][k, 2 * e][d[:12]], 5).tolist()); print("  B", np.round(b["obs0"][k, 2 * e][d[:12]], 5).tolist())
print("ball feats A step t", np.round(a["obs0"][t, 2 * e][:9], 4).tolist()); print("ball feats A step t+1", np.round(a["obs0"][t + 1, 2 * e][:9], 4).tolist()); print("ball feats B step t+1", np.round(b["obs0"][t + 1, 2 * e][:9], 4).tolist())
