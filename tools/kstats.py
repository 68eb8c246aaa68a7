import csv, sys
for r in csv.reader(open(sys.argv[1])):
    print(r[0][:44].ljust(44), r[1].rjust(6), r[3][:9].rjust(10), r[4][:6].rjust(7))
