import mpmath as mp, numpy as np, struct
mp.mp.dps = 60
def hexf(v):
    return float(v).hex()
N = 256
lines = []
for j in range(N):
    t = mp.power(2, mp.mpf(j)/N)
    hi = float(t)
    lo = float(t - mp.mpf(hi))
    lines.append("    {%s, %s}," % (float(hi).hex(), float(lo).hex()))
open("exp2_256.inc", "w").write("\n".join(lines) + "\n")
print(lines[0]); print(lines[1]); print(lines[255])
c = mp.log(2)/N
hi = float(c); lo = float(c - mp.mpf(hi))
print("ln2/256 hi %.17e lo %.17e  inv %.17e" % (hi, lo, float(N/mp.log(2))))
print("ln 40 = %.17g" % float(mp.log(40)))
