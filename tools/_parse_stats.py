import csv,sys
for r in csv.reader(open(sys.argv[1])):
    if r[0].startswith('ff::k_pair') or r[0].startswith('ff::k_merge'):
        print(f"  {r[0].split('(')[0][:50]:50s} calls {r[1]:>5s} avg {float(r[3])/1e3:7.1f} us")
