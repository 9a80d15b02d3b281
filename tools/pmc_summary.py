#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection CSVs under <dir>/pmc_* per (kernel, counter): pass,kernel,counter,dispatches,mean."""
import csv, glob, os, re, sys
from collections import defaultdict
d = sys.argv[1]
print('pass,kernel,counter,dispatches,mean_per_dispatch')
for p in sorted(glob.glob(os.path.join(d, 'pmc_*'))):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(p, '*', '*counter_collection.csv')):
        for row in csv.DictReader(open(f)):
            name = row.get('Kernel_Name', '')
            if 'dam::' not in name:
                continue
            short = re.sub(r'\(.*', '', name.replace('void ', '').replace('dam::(anonymous namespace)::', '').replace('dam::', ''))
            acc[(short, row['Counter_Name'])].append(float(row['Counter_Value']))
    for (k, c), v in sorted(acc.items()):
        print('%s,"%s",%s,%d,%s' % (os.path.basename(p), k, c, len(v), sum(v) / len(v)))
