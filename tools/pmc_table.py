#!/usr/bin/env python
"""One row per kernel, one column per counter (mean per dispatch) from a rocprofv3 --pmc counter_collection.csv."""
import re
import sys

import pandas as pd

df = pd.read_csv(sys.argv[1])
df["k"] = df["Kernel_Name"].map(lambda s: re.sub(r"\(.*", "", s).replace("void ", "")[:48])
t = df.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
n = df.groupby("k")["Dispatch_Id"].nunique()
t.insert(0, "dispatches", n)
t = t.sort_values(t.columns[1], ascending=False).head(28)
cols = list(t.columns)
print("| kernel | " + " | ".join(cols) + " |")
print("|---|" + "---|" * len(cols))
for k, r in t.iterrows():
    print(f"| `{k}` | " + " | ".join(f"{r[c]:.3g}" if c != "dispatches" else str(int(r[c])) for c in cols) + " |")
print()
