"""Slider declarations and the host->slider value bridge.

Mirrors the reference host's behaviour for the caller sequence:
  * `sliderN:DEF<MIN,MAX,STEP{choices}:shape>Label`, optional `sliderN:var=DEF<...>` alias
    -- src/JSFXJuceProcessor.cpp:706-931 (min/max/step/default are parsed to *float32*)
  * value pushed to st.sliders[] = clamp + `min + llround((v-min)/step)*step` + clamp
    -- src/JSFXJuceProcessor.cpp:5556-5596
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field

import numpy as np

_SLIDER_RE = re.compile(r"^\s*slider\s*([0-9]{1,2})\s*:\s*([^<\r\n;]+)\s*(?:<\s*([^>]+)\s*>)?\s*(.*)$")
_NUM_PREFIX_RE = re.compile(r"^\s*[+-]?(?:(?:\d+\.?\d*(?:[eE][+-]?\d+)?)|(?:\.\d+(?:[eE][+-]?\d+)?)|inf(?:inity)?|nan)", re.I)


def _f32(x: float) -> float:
    return float(np.float32(x))


def _strtod_prefix(tok: str):
    """C strtod() semantics: parse the longest numeric prefix, None if there is none."""
    m = _NUM_PREFIX_RE.match(tok)
    if not m:
        return None
    try:
        return float(m.group(0))
    except ValueError:
        return None


def _split_top_level_commas(s: str):
    parts, cur, depth = [], "", 0
    for ch in s:
        if ch == "{":
            depth += 1
        elif ch == "}" and depth > 0:
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    parts.append(cur.strip())
    return parts


@dataclass
class SliderDecl:
    index0: int
    default: float = 0.0
    vmin: float = 0.0
    vmax: float = 1.0
    step: float = 0.001
    var_name: str = ""
    is_choice: bool = False
    is_string: bool = False
    choices: list = field(default_factory=list)
    shape: str = "linear"
    label: str = ""
    hidden: bool = False

    def to_slider_value(self, host_value: float) -> float:
        """hostParameterToJsfxSliderValue() (src/JSFXJuceProcessor.cpp:5556-5596): what lands in st.sliders[index0]. The host's
        parameter value is a float (`std::atomic<float>* paramAtomics`, :9792; `(double) v->load()`, :5566), min / max / step
        are floats widened to double, the arithmetic is double."""
        v = _f32(host_value)
        if self.is_choice:
            v = self.vmin + float(_llround(v)) * self.step
        v = min(max(v, self.vmin), self.vmax)
        if not self.is_choice and self.step > 0.0:
            q = _llround((v - self.vmin) / self.step)
            v = self.vmin + q * self.step
            v = min(max(v, self.vmin), self.vmax)
        return v


def _llround(x: float) -> int:
    if math.isnan(x) or math.isinf(x):
        return 0
    return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def parse_slider_decls(jsfx_text: str):
    """Return {index0: SliderDecl}; first declaration of an index wins (reference sorts + uniques)."""
    out = {}
    for line in re.split(r"[\r\n]+", jsfx_text):
        m = _SLIDER_RE.match(line)
        if not m:
            continue
        n = int(m.group(1))
        if n < 1 or n > 64:
            continue
        d = SliderDecl(index0=n - 1)
        def_full = m.group(2).strip()
        var_tok, def_tok = "", def_full
        eq = def_full.rfind("=")
        if eq >= 0:
            var_tok, def_tok = def_full[:eq].strip(), def_full[eq + 1:].strip()
        dv = _strtod_prefix(def_tok)
        d.default = _f32(dv) if dv is not None else 0.0
        d.var_name = var_tok
        rng = m.group(3)
        if rng is not None and rng.strip().lower() in ("string", "str", "text"):
            d.is_string = True
        if not d.is_string and d.var_name.startswith("#"):
            d.is_string = True
        if not d.is_string and rng is not None:
            parts = _split_top_level_commas(rng)
            vmin, vmax, vstep = 0.0, 1.0, _f32(0.001)
            if len(parts) >= 2:
                a, b = _strtod_prefix(parts[0]), _strtod_prefix(parts[1])
                vmin = _f32(a) if a is not None else 0.0
                vmax = _f32(b) if b is not None else 1.0
            if len(parts) >= 3:
                tok = parts[2]
                br = tok.find("{")
                if br >= 0:
                    cl = tok.find("}", br + 1)
                    if cl >= 0:
                        labels = [t.strip() for t in tok[br + 1:cl].split(",")]
                        labels = [t for t in labels if t]
                        if labels:
                            d.choices, d.is_choice = labels, True
                    tok = tok[:br].strip()
                if ":" in tok:
                    tag = tok.split(":", 1)[1].strip()
                    tok = tok.split(":", 1)[0].strip()
                    base = tag.split("=", 1)[0].strip()
                    if base in ("log", "sqr"):
                        d.shape = base
                if tok == "":
                    vstep = 1.0
                else:
                    s = _strtod_prefix(tok)
                    vstep = _f32(s) if s is not None else 1.0
            if vmax < vmin:
                vmin, vmax = vmax, vmin
            d.vmin, d.vmax = vmin, vmax
            d.step = vstep if vstep > 0.0 else _f32(0.001)
            d.default = min(max(d.default, d.vmin), d.vmax)
        label = m.group(4).strip()
        if label.startswith("-"):
            d.hidden = True
            label = label[1:].lstrip()
        d.label = label or f"Slider {n}"
        out.setdefault(d.index0, d)
    return out


def default_slider_values(decls) -> np.ndarray:
    """64 slider values as pushParamsToStateSliders() would write them for untouched host params."""
    v = np.zeros(64, dtype=np.float64)
    for i, d in decls.items():
        if d.is_string:
            continue
        host = d.default
        if d.is_choice:  # host param holds the choice index
            host = (d.default - d.vmin) / d.step if d.step > 0 else 0.0
        v[i] = d.to_slider_value(host)
    return v


def slider_aliases(decls) -> dict:
    return {i: d.var_name for i, d in decls.items() if d.var_name and not d.var_name.startswith("#")}
