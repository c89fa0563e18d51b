#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc CSV output (counter_collection.csv): counter value per dispatch."""
import re
import sys

import pandas as pd


def main(path, top=24):
    df = pd.read_csv(path)
    df["k"] = df["Kernel_Name"].map(lambda s: re.sub(r"\(.*", "", s).replace("void ", "")[:70])
    g = df.groupby(["k", "Counter_Name"])["Counter_Value"].agg(["count", "mean"]).reset_index()
    g["total"] = g["count"] * g["mean"]
    for c in g["Counter_Name"].unique():
        s = g[g["Counter_Name"] == c].sort_values("total", ascending=False).head(top)
        print(f"## {c}")
        print("| kernel | dispatches | mean per dispatch |\n|---|---|---|")
        for _, r in s.iterrows():
            print(f"| `{r['k']}` | {int(r['count'])} | {r['mean']:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
