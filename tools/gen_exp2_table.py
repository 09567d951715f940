import mpmath as mp, numpy as np, struct
mp.mp.dps = 60
def hexf(v):
    return float(v).hex()
N = 256
lines = []
for j in range(N):
    t = mp.power(2, mp.mpf(j)/N)
    lines.append("    %s," % float(t).hex())        # 2^(j/256) rounded to nearest (rounds 2-5: a hi/lo pair)
open("exp2_256.inc", "w").write("\n".join(lines) + "\n")
print(lines[0]); print(lines[1]); print(lines[255])
c = mp.log(2)/N
hi = float(c); lo = float(c - mp.mpf(hi))
print("ln2/256 hi %.17e lo %.17e  inv %.17e" % (hi, lo, float(N/mp.log(2))))
print("ln 40 = %.17g" % float(mp.log(40)))
