set -e
L=secure-video-steganography-using-ecc-and-dct_amd/lib/variants   # the experiments library (make -C csrc exp): the product library reads no environment
for cfg in "--n-ac 3 --delta 8" "--n-ac 10 --delta 20" "--n-ac 7 --delta 8" "--n-ac 20 --delta 12 --frames 300"; do
  python tools/ab_bench.py $cfg --rounds 7 --env-sweep SVS_EMBED_WG_PER_CU=0,8,6,5,4,3 $L/libsvsdct_exp.so | grep -E 'frames|embed med' | sed 's/ | extract.*//'
  python tools/ab_bench.py $cfg --rounds 7 --env-sweep SVS_EXTRACT_WG_PER_CU=0,8,6,5,4,3 $L/libsvsdct_exp.so | grep -E 'extract med' | sed 's/embed med.*| extract/extract/;s/ | [0-9a-f]* roundtrip.*//'
done
python tools/ab_bench.py --mode exact --n-ac 3 --delta 8 --rounds 5 --env-sweep SVS_EMBED_WG_PER_CU=0,3,2 $L/libsvsdct_exp.so | grep -E 'frames|embed med' | sed 's/ | extract.*//'
