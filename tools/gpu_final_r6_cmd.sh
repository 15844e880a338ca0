export COMMIT=$(cat tools/.evidence_commit 2>/dev/null || echo HEAD) OUT=r06final WITH_TESTS=1 WITH_HUNTS=1
export WL="cfg2:config2/fft/float32/3000/w120/m120/n1:|cfg2u8:config2/fft/uint8/3000/w120/m120/n1:--sample-type uint8|hard:config2/fft/float32/3000/w120/m120/n1/hard0.05/off7.25:--hard-frac 0.05|cc:config2/fft/float32/3000/w120/m120/n1/ccoeff_normed:--method ccoeff_normed|cfg1:config1/fft/float32/1000/w60/m45/n1:--config 1|cfg4:config4/fft/float32/5000/w120/m240/n1:--config 4"
export EXTRA="lanes1:|snr12:--snr 12|snr6:--snr 6|snr0:--snr 0|unrelated:--unrelated|encode:--source encode|encodeu8:--source encode --sample-type uint8|dub:--source dub|partial:--source partial|stat20:|stat6:--snr 6|whole:--exclusion whole|never:--exclusion never"
bash tools/gpu_final_r6.sh
