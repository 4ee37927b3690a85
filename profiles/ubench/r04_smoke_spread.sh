echo "== f32 SPREAD_NORECT=dgrad"; SPREAD_NORECT=dgrad timeout -k 10 500 python profiles/ubench/smoke_spread.py 5 f32 2>&1 | grep "^run" | cut -c1-330
echo "== f32 SPREAD_NORECT=fwd"; SPREAD_NORECT=fwd timeout -k 10 500 python profiles/ubench/smoke_spread.py 5 f32 2>&1 | grep "^run" | cut -c1-330
