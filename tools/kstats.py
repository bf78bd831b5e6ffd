#!/usr/bin/env python
"""print name / calls / average us of a rocprofv3 kernel_stats.csv (names contain commas: a real CSV reader)"""
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    print("%-90s n=%6s avg %9.2f us" % (r["Name"].replace("(anonymous namespace)::", "")[:90], r["Calls"], float(r["AverageNs"]) / 1e3))
