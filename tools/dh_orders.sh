# same-box comparison of dhconv_strip variants (tools/mkvar.sh / mkone.sh) and unit orders (ACE_DH_ORDER, measurement builds): usage dh_orders.sh lib:order ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "$@"; do lib=${cfg%%:*}; o=${cfg##*:}
  ACE_DH_ORDER=$o bash tools/kdur2.sh ab_${lib}_o${o}_$rep $GRAFT_REPO_ROOT/exp/libexp_$lib.so > /dev/null 2>&1
  echo "== $lib order $o (rep $rep): $(grep -h '^steps/s' gpurun_out/kdur_ab_${lib}_o${o}_$rep.txt)"; grep -h dhconv_strip gpurun_out/kdur_ab_${lib}_o${o}_$rep.txt | cut -c1-30,67-140
done; done
