"""Generator of the fx_fuzz*.jsfx fixture scripts (repo-authored, deterministic from the seed).

Each script is a random program over the constructs both lowerings of the reference agree on (SURVEY §8 a-2 addendum lists
the AOT-vs-EEL2 deltas that are avoided here: no == / !=, conditions are exact 0/1, % and the bit operators only see
non-negative integers below 2^20, loop counts are small non-negative integers, mem[] indices are integers in [0, 96),
magnitudes stay far from the denormal filter). It exercises: arithmetic, ordered comparisons, ?: with and without else,
&& || !, | & << >> %, |= &= ~=, ^ (pow), min/max/abs/floor/ceil/sqrt/sin/cos/exp/log/atan2/sign/sqr/invsqrt, loop(), while(),
mem[] loads/stores with computed indices, compound assignments, user functions with parameters and persistent local()
variables called from @init, @block and @sample, spl0/spl1 I/O and a slider.

    python tests/fixtures/make_fuzz.py          # rewrites tests/fixtures/fuzz0.jsfx ... fuzz5.jsfx

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


if __name__ == "__main__":
    main()
