"""DSC network files in the dialect the reference's loader accepts
(``bayesian/serializer/dsc.hpp:33-227``, the only Boost-free on-disk format of the reference).

    belief network "name"
    node NAME
    {
      type: discrete[K] = { ... };
    }
    probability(NAME | P1, P2)
    {
      (s1, s2): p0, p1, ...;
    }
    probability(ROOT)
    {
      p0, p1, ...;
    }

Lines are trimmed, empty lines and ``//`` comment lines skipped.  Node index = order of the
``node`` blocks (the position in ``vertex_list()``; the reference's loader never sets
``vertex_t::id``).  The parents listed after ``|`` name the columns of the ``(s1, s2)`` row keys;
the flat model stores parents in ascending node index, so rows are re-ordered accordingly.
"""
from __future__ import annotations

import re

import numpy as np

from .flat import FlatModel


class DscError(ValueError):
    pass


def parse_dsc(text: str):
    """Returns (FlatModel, names).  Raises DscError on anything the reference would mis-parse."""
    lines = [ln.strip() for ln in text.replace("\r", "").split("\n")]
    lines = [ln for ln in lines if ln and not ln.startswith("//")]
    names, arity, index = [], [], {}
    tables = {}  # node index -> (listed parent indices, {state tuple: row})
    name = ""
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("belief network"):
            m = re.match(r'belief network\s+"(.*)"', ln)
            name = m.group(1) if m else ""
            i += 1
        elif ln.startswith("node "):
            nm = ln[5:].strip()
            if nm in index:
                raise DscError(f"node {nm} declared twice")
            if i + 1 >= len(lines) or lines[i + 1] != "{":
                raise DscError(f"node {nm}: '{{' expected on its own line")
            i += 2
            k = None
            while i < len(lines) and lines[i] != "}":
                m = re.match(r"type\s*:\s*discrete\s*\[\s*(\d+)\s*\]", lines[i])
                if m:
                    k = int(m.group(1))
                i += 1
            if k is None or k < 1:
                raise DscError(f"node {nm}: no 'type: discrete[K]' line")
            index[nm] = len(names)
            names.append(nm)
            arity.append(k)
            i += 1
        elif ln.startswith("probability"):
            m = re.match(r"probability\s*\(\s*([^|)]+?)\s*(?:\|\s*(.*?))?\s*\)\s*$", ln)
            if not m:
                raise DscError(f"cannot parse '{ln}'")
            target = m.group(1).strip()
            parents = [p.strip() for p in m.group(2).split(",")] if m.group(2) else []
            for p in [target] + parents:
                if p not in index:
                    raise DscError(f"unknown node '{p}' in '{ln}'")
            if i + 1 >= len(lines) or lines[i + 1] != "{":
                raise DscError(f"probability({target}): '{{' expected on its own line")
            i += 2
            rows = {}
            while i < len(lines) and lines[i] != "}":
                body = lines[i].rstrip(";").strip()
                if parents:
                    m2 = re.match(r"\(\s*(.*?)\s*\)\s*:\s*(.*)$", body)
                    if not m2:
                        raise DscError(f"probability({target}): bad row '{lines[i]}'")
                    key = tuple(int(x) for x in m2.group(1).split(","))
                    vals = m2.group(2)
                else:
                    key, vals = (), body
                if len(key) != len(parents):
                    raise DscError(f"probability({target}): row key {key} does not name {len(parents)} parents")
                rows[key] = [float(x) for x in vals.split(",")]
                i += 1
            tables[index[target]] = ([index[p] for p in parents], rows)
            i += 1
        else:
            i += 1

    n = len(names)
    k = np.asarray(arity, dtype=np.int32)
    in_ptr = np.zeros(n + 1, dtype=np.int32)
    in_idx, cpt, cpt_off = [], [], np.zeros(n + 1, dtype=np.int64)
    for v in range(n):
        if v not in tables:
            raise DscError(f"node {names[v]} has no probability block")
        listed, rows = tables[v]
        if len(set(listed)) != len(listed) or v in listed:
            raise DscError(f"probability({names[v]}): duplicate or self parent")
        order = sorted(range(len(listed)), key=lambda j: listed[j])     # ascending node index
        parents = [listed[j] for j in order]
        in_idx.extend(parents)
        in_ptr[v + 1] = len(in_idx)
        radix = [int(k[p]) for p in parents]
        state = [0] * len(parents)
        total = int(np.prod(radix)) if radix else 1
        for _ in range(total):
            key = [0] * len(listed)
            for pos, j in enumerate(order):
                key[j] = state[pos]
            row = rows.get(tuple(key))
            if row is None or len(row) != int(k[v]):
                raise DscError(f"probability({names[v]}): row {tuple(key)} missing or not {int(k[v])} long")
            cpt.extend(row)
            for pos in range(len(parents) - 1, -1, -1):               # first parent most significant
                state[pos] += 1
                if state[pos] < radix[pos]:
                    break
                state[pos] = 0
        cpt_off[v + 1] = len(cpt)
    model = FlatModel(k, in_ptr, np.asarray(in_idx, dtype=np.int32), cpt_off, np.asarray(cpt, dtype=np.float64),
                      name=name or "dsc")
    model.validate()
    return model, names


def load_dsc(path: str):
    with open(path) as f:
        return parse_dsc(f.read())
