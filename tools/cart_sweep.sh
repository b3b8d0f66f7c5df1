# CartPole C3 kernel: roots per wave x replication sweep (tools/micro_cartpole.py; best of 40 launches, host arrays in / out)
export MI355PLAN_NO_TORCH=1
echo "default: $(python tools/micro_cartpole.py 4096 | tail -1)"
echo "default, IEEE divisions: $(MP_CART_FASTDIV=0 python tools/micro_cartpole.py 4096 | tail -1)"
for cfg in "16 0" "4 2" "4 4" "2 5" "8 3" "8 2" "16 2" "1 6"; do
  set -- $cfg
  echo "lanes $1 rep $2: $(MP_UCT_LANES=$1 MP_UCT_CART_REP=$2 python tools/micro_cartpole.py 4096 | tail -1)"
done
