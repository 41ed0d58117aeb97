"""Generator of the fx_fuzz*.jsfx fixture scripts (repo-authored, deterministic from the seed).

Each script is a random program over the constructs both lowerings of the reference agree on (SURVEY §8 a-2 addendum lists
the AOT-vs-EEL2 deltas that are avoided here: no == / !=, conditions are exact 0/1, % and the bit operators only see
non-negative integers below 2^20, loop counts are small non-negative integers, mem[] indices are integers in [0, 96),
magnitudes stay far from the denormal filter). It exercises: arithmetic, ordered comparisons, ?: with and without else,
&& || !, | & << >> %, |= &= ~=, ^ (pow), min/max/abs/floor/ceil/sqrt/sin/cos/exp/log/atan2/sign/sqr/invsqrt, loop(), while(),
mem[] loads/stores with computed indices, compound assignments, user functions with parameters and persistent local()
variables called from @init, @block and @sample, spl0/spl1 I/O and a slider.

    python tests/fixtures/make_fuzz.py          # rewrites tests/fixtures/fuzz0.jsfx ... fuzz49.jsfx

fuzz0 .. fuzz5 are round 1's (unchanged: same seeds, same text). fuzz6 .. fuzz49 (round 4) add what the time-parallel lowering
has to get right beyond plain dataflow: ring buffers written at a moving position and read at fixed and data-dependent lags,
a feedback echo through a ring (lags above and below a chunk's 64 frames), stores into a ring under a per-frame condition,
counted loops over per-trip cells (band states) under conditions, counters that wrap with a mask / a modulo / a compare, a
rare heavy branch (a loop run every few hundred frames), and user functions with instance state called at several sites.

The reference VM's outputs for them are frozen by tests/golden/make_golden.py like for every other fixture leaf.
"""
import random
from pathlib import Path

HERE = Path(__file__).resolve().parent
VARS = ["a", "b", "c", "d", "e", "f", "g", "h"]


class Gen:
    def __init__(self, seed):
        self.r = random.Random(seed)
        self.depth = 0

    def num(self):
        return self.r.choice(["0.5", "1", "2", "3", "0.25", "1.5", "7", "0.125", "10", "0.75", "$pi", "4"])

    def var(self):
        return self.r.choice(VARS)

    def small_int(self, e):            # non-negative integer below 2^20 from an arbitrary value
        return f"(floor(abs({e})) & 1048575)"

    def index(self):                   # integer in [0, 96)
        return f"(floor(abs({self.expr(2)})) % 96)"

    def cond(self, d):
        k = self.r.randrange(6)
        a, b = self.expr(d - 1), self.expr(d - 1)
        if k == 0: return f"({a} < {b})"
        if k == 1: return f"({a} > {b})"
        if k == 2: return f"({a} <= {b})"
        if k == 3: return f"({a} >= {b})"
        if k == 4: return f"(({a} < {b}) && ({self.expr(d - 1)} > 0.5))"
        return f"(!({a} < {b}) || ({self.expr(d - 1)} > 1))"

    def expr(self, d=3):
        if d <= 0 or self.r.random() < 0.25:
            k = self.r.randrange(5)
            if k == 0: return self.num()
            if k == 1: return f"mem[{self.r.randrange(96)}]"
            if k == 2: return "spl0" if self.r.random() < 0.5 else "spl1"
            return self.var()
        k = self.r.randrange(22)
        a, b = self.expr(d - 1), self.expr(d - 1)
        if k == 0: return f"({a} + {b})"
        if k == 1: return f"({a} - {b})"
        if k == 2: return f"({a} * {b})"
        if k == 3: return f"({a} / (1 + abs({b})))"
        if k == 4: return f"min({a}, {b})"
        if k == 5: return f"max({a}, {b})"
        if k == 6: return f"abs({a})"
        if k == 7: return f"sin({a})"
        if k == 8: return f"cos({a})"
        if k == 9: return f"sqrt(abs({a}))"
        if k == 10: return f"exp(min({a}, 3))"
        if k == 11: return f"log(1 + abs({a}))"
        if k == 12: return f"floor({a})"
        if k == 13: return f"ceil({a})"
        if k == 14: return f"({self.cond(d)} ? {a} : {b})"
        if k == 15: return f"({self.small_int(a)} | {self.small_int(b)})"
        if k == 16: return f"({self.small_int(a)} & {self.small_int(b)})"
        if k == 17: return f"({self.small_int(a)} % (1 + ({self.small_int(b)} & 15)))"
        if k == 18: return f"((1 + abs({a})) ^ (min(abs({b}), 2)))"
        if k == 19: return f"mem[{self.index()}]"
        if k == 20: return f"atan2({a}, 1 + abs({b}))"
        return self.r.choice([f"sqr({a})", f"sign({a})", f"invsqrt(1 + abs({a}))", f"(({self.small_int(a)} << 3) >> 1)"])

    def clamp(self, e):
        return f"max(-50, min(50, {e}))"

    def stmt(self, d=2, in_fn=False):
        k = self.r.randrange(12)
        v = self.var()
        if k <= 2: return f"{v} = {self.clamp(self.expr(3))};"
        if k == 3:
            if self.r.random() < 0.3:   # the integer compound assignments (~= is xor; there is no binary ~ in the AOT grammar)
                return f"{v} = {self.small_int(v)}; {v} {self.r.choice(['|=', '&=', '~='])} {self.small_int(self.expr(2))}; {v} = {self.clamp(v)};"
            return f"{v} {self.r.choice(['+=', '-=', '*=', '/='])} {self.clamp('1 + abs(' + self.expr(2) + ')')}; {v} = {self.clamp(v)};"
        if k == 4: return f"mem[{self.index()}] = {self.clamp(self.expr(3))};"
        if k == 5 and d > 0:
            n = self.r.randrange(1, 6)
            return f"i = 0; loop({n}, mem[(i + {self.r.randrange(90)}) % 96] = {self.clamp(self.expr(2))}; i += 1; );"
        if k == 6 and d > 0:
            n = self.r.randrange(1, 5)
            return f"j = 0; while (j < {n}) ( {v} = {self.clamp(self.expr(2))}; j += 1; );"
        if k == 7 and d > 0:
            return f"{self.cond(2)} ? ( {self.stmt(d - 1, in_fn)} ) : ( {self.stmt(d - 1, in_fn)} );"
        if k == 8 and d > 0:
            return f"{self.cond(2)} ? ( {self.stmt(d - 1, in_fn)} );"
        if k == 9 and not in_fn: return f"{v} = {self.clamp('f1(' + self.expr(2) + ', ' + self.expr(2) + ')')};"
        if k == 10 and not in_fn: return f"{v} = {self.clamp('f2(' + self.expr(2) + ')')};"
        return f"{v} = {self.clamp(self.expr(2))};"

    # ---- round 4: memory idioms (all addresses >= 128: the random mem[] traffic of stmt() stays in [0, 96)) -------------------
    def ring_stmts(self):
        """A few statements from the delay-line / band-loop / counter families; each family uses its own region of mem[]."""
        r, out = self.r, []
        fams = r.sample(["ring", "echo", "condring", "bands", "wrap", "rare", "inst"], r.randrange(3, 6))
        for fam in fams:
            v = self.var()
            if fam == "ring":                 # written every frame at a masked free-running counter; fixed and signal-dependent lags
                lag = r.choice([1, 3, 17, 63, 64, 65, 100])
                out.append(f"mem[256 + (rw & 127)] = {self.clamp(self.expr(2))};")
                out.append(f"{v} = mem[256 + ((rw - {lag}) & 127)] * 0.5 + mem[256 + ((rw - 1 - (floor(abs({self.var()})) & 31)) & 127)] * 0.25;")
                out.append("rw += 1;")
            elif fam == "echo":               # feedback through a ring: the stored value depends on what is read from it
                lag = r.choice([20, 40, 64, 90, 200])
                out.append(f"eo = mem[512 + ((ew - {lag}) & 255)];")
                out.append(f"mem[512 + ew] = {self.clamp('spl0 + eo * 0.5 + ' + self.var() + ' * 0.01')}; ew = (ew + 1) & 255;")
                out.append(f"{v} = {self.clamp(v + ' * 0.5 + eo')};")
            elif fam == "condring":           # a store under a per-frame condition into a buffer nothing reads on the audio path
                out.append(f"{self.cond(1)} ? ( mem[1024 + cw] = {self.clamp(self.expr(2))}; cw += 1; cw >= 50 ? cw = 0; );")
            elif fam == "bands":              # per-trip cells: a bank of one-poles, every second one only under a condition
                n = r.randrange(3, 7)
                cnd = self.cond(1)
                out.append(f"bi = 0; bs = 0; loop({n}, bz = mem[1200 + bi]; bz += (0.1 + bi * 0.05) * ({self.clamp(self.expr(1))} - bz); "
                           f"(bi & 1) ? ( {cnd} ? mem[1200 + bi] = bz; ) : ( mem[1200 + bi] = bz; ); bs += mem[1300 + bi] * bz; bi += 1; );")
                out.append(f"{v} = {self.clamp('bs')};")
            elif fam == "wrap":               # counters that wrap: modulo, mask, compare
                out.append(f"wa = (wa + {r.choice([1, 3, 5])}) % {r.choice([37, 64, 100])}; wb = (wb + 1) & {r.choice([31, 63, 255])}; "
                           f"wc += 1; wc >= {r.choice([10, 64, 77])} ? wc = 0;")
                out.append(f"{v} = {self.clamp(v + ' * 0.5 + (wa + wb - wc) * 0.01')};")
            elif fam == "rare":               # a heavy branch every few hundred frames
                per = r.choice([97, 128, 300])
                out.append(f"rc += 1; rc >= {per} ? ( rc = 0; ri = 0; loop(24, mem[1400 + ri] = mem[1400 + ri] * 0.9 + {self.var()} * 0.1 + ri; ri += 1; ); {v} = {self.clamp('mem[1400 + 7]')}; );")
            else:                             # instance state, two call sites
                out.append(f"{v} = {self.clamp('pa.follow(' + self.expr(1) + ') - pb.follow(' + self.expr(1) + ')')};")
        r.shuffle(out) if False else None       # (statements of one family stay in order)
        return out

    def program2(self, idx):
        L = [f"desc:random semantics fixture {idx}, memory idioms (generated by tests/fixtures/make_fuzz.py; repo-authored)",
             "slider1:3<0,10,0.5>amount", "", "@init"]
        for i, v in enumerate(VARS):
            L.append(f"{v} = {0.25 * (i + 1) * (1 if i % 2 else -1)};")
        L.append("k = 0; loop(96, mem[k] = sin(k * 0.37) * 4; k += 1; );")
        L.append("k = 0; loop(8, mem[1300 + k] = 1 / (1 + k); k += 1; );")
        L.append("rw = 0; ew = 0; cw = 0; wa = 0; wb = 0; wc = 0; rc = 0;")
        L.append("function f1(x, y) local(acc, n) ( acc = max(-50, min(50, acc * 0.5 + x)); n += 1; " +
                 f"{self.clamp(self.expr(2).replace('spl0', 'x').replace('spl1', 'y'))} + acc * 0.125 + (n & 3); );")
        L.append("function f2(x) local(z) ( z = max(-50, min(50, z + x * 0.25)); " + "x > 0 ? z : 0 - z; );")
        L.append("function follow(x) instance(env) ( env += (abs(x) > env ? 0.3 : 0.01) * (abs(x) - env); env; );")
        L += ["", "@slider", f"h = {self.clamp('slider1 * 0.5 + ' + self.expr(2))};"]
        if self.r.random() < 0.6:
            L += ["", "@block"]
            for _ in range(2):
                L.append(self.stmt(1))
        L += ["", "@sample"]
        body = [self.stmt(2) for _ in range(self.r.randrange(3, 7))]
        ring = self.ring_stmts()
        at = self.r.randrange(0, len(body) + 1)
        L += body[:at] + ring + body[at:]
        L.append(f"spl0 = max(-4, min(4, {self.expr(3)} * 0.1));")
        L.append(f"spl1 = max(-4, min(4, ({self.var()} + {self.var()}) * 0.05 + spl1 * 0.5));")
        return "\n".join(L) + "\n"

    def program(self, idx):
        L = [f"desc:random semantics fixture {idx} (generated by tests/fixtures/make_fuzz.py; repo-authored)",
             "slider1:3<0,10,0.5>amount", "", "@init"]
        for i, v in enumerate(VARS):
            L.append(f"{v} = {0.25 * (i + 1) * (1 if i % 2 else -1)};")
        L.append("k = 0; loop(96, mem[k] = sin(k * 0.37) * 4; k += 1; );")
        L.append("function f1(x, y) local(acc, n) ( acc = max(-50, min(50, acc * 0.5 + x)); n += 1; " +
                 f"{self.clamp(self.expr(2).replace('spl0', 'x').replace('spl1', 'y'))} + acc * 0.125 + (n & 3); );")
        L.append("function f2(x) local(z) ( z = max(-50, min(50, z + x * 0.25)); " + "x > 0 ? z : 0 - z; );")
        for _ in range(4):
            L.append(self.stmt(2))
        L += ["", "@slider", f"h = {self.clamp('slider1 * 0.5 + ' + self.expr(2))};", "", "@block"]
        for _ in range(3):
            L.append(self.stmt(1))
        L += ["", "@sample"]
        for _ in range(self.r.randrange(8, 14)):
            L.append(self.stmt(2))
        L.append(f"spl0 = max(-4, min(4, {self.expr(3)} * 0.1));")
        L.append(f"spl1 = max(-4, min(4, ({self.var()} + {self.var()}) * 0.05 + spl1 * 0.5));")
        return "\n".join(L) + "\n"


def main():
    for idx in range(6):
        (HERE / f"fuzz{idx}.jsfx").write_text(Gen(1000 + idx).program(idx))
        print("wrote", f"fuzz{idx}.jsfx")
    for idx in range(6, 50):
        (HERE / f"fuzz{idx}.jsfx").write_text(Gen(1000 + idx).program2(idx))
        print("wrote", f"fuzz{idx}.jsfx")


if __name__ == "__main__":
    main()
