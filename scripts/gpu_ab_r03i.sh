V="0:0,6:0,8:0,10:0,14:0,16:0,20:0"
echo "== budget 6(default),3,4,5,7,8,10: sphere 100k"; AB_VARIANTS=$V AB_ROUNDS=3 python scripts/ab_tuning.py 2>&1 | tail -7
echo "== sphere 1M"; AB_VARIANTS=$V AB_ROUNDS=3 AB_TRIS=1000000 AB_SIZE=4096 AB_SPP=4 python scripts/ab_tuning.py 2>&1 | tail -7
echo "== random"; AB_VARIANTS=$V AB_ROUNDS=2 AB_SCENE=random AB_SPP=16 python scripts/ab_tuning.py 2>&1 | tail -7
echo "== cornell 36 tris 512^2"; AB_VARIANTS=$V AB_TRIS=30 AB_SIZE=512 python scripts/ab_tuning.py 2>&1 | tail -7
