cd $GRAFT_REPO_ROOT
O=gpurun_out/g4a; mkdir -p $O
timeout 120 tools/store_probe > $O/store_probe.txt 2>&1; grep "grid 256" $O/store_probe.txt | grep "delay     0\|delay 45000 stagger 0"
VARIANTS=0 G4=0,1,4,5,8 REPS=3 SHAPES="qkv:20800:2304:768,proj:20800:768:768,fc2:20800:768:3072,sq4096:4096:4096:4096" timeout 600 python tools/g8_lab.py 2>&1 | tee $O/lab1.txt | tail -n 14
VARIANTS=100 G4=100 REPS=2 SHAPES="fc1:20800:3072:768" timeout 300 python tools/g8_lab.py 2>&1 | tee $O/lab2.txt | tail -n 3
VARIANTS=0 G4=200 REPS=2 SHAPES="fc2:20800:768:3072,vits:8192:384:384,vits2:8192:1152:384,vitl:5840:1024:1024" timeout 300 python tools/g8_lab.py 2>&1 | tee $O/lab3.txt | tail -n 9
