# Build A/B variants of libnepmi.so: bash profiles/ab_variants.sh name1:"-DX=1 -DY=2" name2:"..."
# -> gpumd_amd/lib/variants/libnepmi_<name>.so (git-ignored; they travel with the gpurun snapshot)
cd /root/repo/gpumd_amd/csrc
mkdir -p ../lib/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed -Wl,-Bsymbolic -Wl,-rpath,/opt/rocm/lib $flags \
      -o ../lib/variants/libnepmi_$name.so engine.hip nep_model.cpp transport_tcp.cpp -ldl 2>&1 | grep -E "error" ) &
done
wait
ls -la ../lib/variants/
