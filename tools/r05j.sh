for i in n1 n2 n3 n4; do echo "== variant $i"; for w in 1024; do ./tools/probes/infer_probe_$i $w 50 | tail -1; done; done
