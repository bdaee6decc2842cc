for i in 1 2; do
for v in prev ""; do
python tools/dev_fit_once.py 4096 - 200 $v
python tools/dev_fit_x.py 1024 0 50 $v
python tools/dev_fit_x.py 4096 0 30 $v
done; done
