#!/bin/bash
# Samples rocm-smi (power, sclk, temperature) twice a second while a command runs: tools/power_sample.sh <out.txt> <command ...>
OUT=$1; shift
( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor junction\)|mclk" | tr '\n' ' '; echo; sleep 0.5; done ) > $OUT.raw 2>&1 &
SP=$!
"$@"
kill $SP
python3 - <<PY
import re, statistics
pw=[]; sc=[]; tj=[]
for l in open("$OUT.raw"):
    m=re.search(r"Power \(W\): ([\d.]+)", l); 
    if m: pw.append(float(m.group(1)))
    m=re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", l)
    if m: sc.append(float(m.group(1)))
    m=re.search(r"junction\) \(C\): ([\d.]+)", l)
    if m: tj.append(float(m.group(1)))
def s(x): return "n=%d min %.0f median %.0f max %.0f" % (len(x), min(x), statistics.median(x), max(x)) if x else "none"
open("$OUT","w").write("power W: %s\nsclk MHz: %s\njunction C: %s\n" % (s(pw), s(sc), s(tj)))
print(open("$OUT").read())
PY
