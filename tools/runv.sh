for c in 1 8 16 32 64 128; do python tools/variants.py one build/variants/base.so 2e7 accum_copies=$c; done
python tools/variants.py one build/variants/nodep.so 2e7
python tools/variants.py one build/variants/nodep4.so 2e7
