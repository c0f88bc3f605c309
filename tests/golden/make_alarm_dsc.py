#!/usr/bin/env python3
"""Writes tests/golden/alarm_shaped.dsc: a 37-node / 46-arc network with the structure and arities
of the ALARM monitoring network (Beinlich et al. 1989), in the DSC dialect of the reference's
loader (bayesian/serializer/dsc.hpp).  ALARM itself ships neither with the reference nor with this
container, so the CPT VALUES are synthetic (seeded, strictly positive); parity is judged
reference-vs-this-repo on the same file."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd.synth import uniform01  # noqa: E402

NODES = [("HISTORY", 2), ("CVP", 3), ("PCWP", 3), ("HYPOVOLEMIA", 2), ("LVEDVOLUME", 3), ("LVFAILURE", 2),
         ("STROKEVOLUME", 3), ("ERRLOWOUTPUT", 2), ("HRBP", 3), ("HREKG", 3), ("ERRCAUTER", 2), ("HRSAT", 3),
         ("INSUFFANESTH", 2), ("ANAPHYLAXIS", 2), ("TPR", 3), ("EXPCO2", 4), ("KINKEDTUBE", 2), ("MINVOL", 4),
         ("FIO2", 2), ("PVSAT", 3), ("SAO2", 3), ("PAP", 3), ("PULMEMBOLUS", 2), ("SHUNT", 2), ("INTUBATION", 3),
         ("PRESS", 4), ("DISCONNECT", 2), ("MINVOLSET", 3), ("VENTMACH", 4), ("VENTTUBE", 4), ("VENTLUNG", 4),
         ("VENTALV", 4), ("ARTCO2", 3), ("CATECHOL", 2), ("HR", 3), ("CO", 3), ("BP", 3)]
PARENTS = {"HISTORY": ["LVFAILURE"], "CVP": ["LVEDVOLUME"], "PCWP": ["LVEDVOLUME"],
           "LVEDVOLUME": ["HYPOVOLEMIA", "LVFAILURE"], "STROKEVOLUME": ["HYPOVOLEMIA", "LVFAILURE"],
           "HRBP": ["ERRLOWOUTPUT", "HR"], "HREKG": ["ERRCAUTER", "HR"], "HRSAT": ["ERRCAUTER", "HR"],
           "TPR": ["ANAPHYLAXIS"], "EXPCO2": ["ARTCO2", "VENTLUNG"], "MINVOL": ["INTUBATION", "VENTLUNG"],
           "PVSAT": ["FIO2", "VENTALV"], "SAO2": ["PVSAT", "SHUNT"], "PAP": ["PULMEMBOLUS"],
           "SHUNT": ["INTUBATION", "PULMEMBOLUS"], "PRESS": ["INTUBATION", "KINKEDTUBE", "VENTTUBE"],
           "VENTMACH": ["MINVOLSET"], "VENTTUBE": ["DISCONNECT", "VENTMACH"],
           "VENTLUNG": ["INTUBATION", "KINKEDTUBE", "VENTTUBE"], "VENTALV": ["INTUBATION", "VENTLUNG"],
           "ARTCO2": ["VENTALV"], "CATECHOL": ["ARTCO2", "INSUFFANESTH", "SAO2", "TPR"], "HR": ["CATECHOL"],
           "CO": ["HR", "STROKEVOLUME"], "BP": ["CO", "TPR"]}


def main():
    arity = dict(NODES)
    assert len(NODES) == 37 and sum(len(p) for p in PARENTS.values()) == 46
    out = ['belief network "alarm_shaped"']
    for name, k in NODES:
        states = ", ".join(f'"s{i}"' for i in range(k))
        out += [f"node {name}", "{", f"  type: discrete[{k}] = {{ {states} }};", "}"]
    draw = 0
    for name, k in NODES:
        ps = PARENTS.get(name, [])
        out.append(f"probability({name}" + (" | " + ", ".join(ps) if ps else "") + ")")
        out.append("{")
        radix = [arity[p] for p in ps]
        state = [0] * len(ps)
        rows = 1
        for r in radix:
            rows *= r
        for _ in range(rows):
            u = 0.05 + 0.95 * uniform01(1989, draw, k)
            draw += k
            row = u / u.sum()
            txt = ", ".join(repr(float(x)) for x in row) + ";"
            out.append(("  (" + ", ".join(map(str, state)) + "): " + txt) if ps else ("  " + txt))
            for j in range(len(ps) - 1, -1, -1):
                state[j] += 1
                if state[j] < radix[j]:
                    break
                state[j] = 0
        out.append("}")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "alarm_shaped.dsc")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
