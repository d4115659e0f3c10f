import csv, sys
rows = list(csv.DictReader(open(f'gpurun_out/prof_{sys.argv[1]}/r_kernel_stats.csv')))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tot = sum(float(r['TotalDurationNs']) for r in rows)/1e6
print(f'total kernel ms {tot:.1f}  (per step {tot/4:.1f})')
for r in rows[:n]:
    print(f"{r['Name'][:58]:58s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:8.2f} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={float(r['Percentage']):.1f}")
