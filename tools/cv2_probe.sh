#!/bin/bash
# Is there ANY route to the real cv2 (OpenCV's Python module) on this machine?  VERDICT r4 item 2: record the outcome either way.
# Writes gpurun_out/cv2_probe.txt; when cv2 does import, also makes the golden fixture and runs the cross-check tests.
out=gpurun_out/cv2_probe.txt
mkdir -p gpurun_out
{
echo "== $(date -u) host $(hostname) =="
echo "-- import cv2 (python3, conda python if any)"
python3 -c 'import cv2; print("cv2", cv2.__version__)' 2>&1 | tail -1
for py in /opt/conda/bin/python /usr/bin/python3 /usr/local/bin/python3; do [ -x $py ] && { echo "$py:"; $py -c 'import cv2; print("cv2", cv2.__version__)' 2>&1 | tail -1; }; done
echo "-- files named like OpenCV anywhere on the box"
find / -xdev \( -iname 'cv2*.so' -o -iname 'opencv*.whl' -o -iname 'libopencv*' -o -iname 'opencv_python*' \) 2>/dev/null | head -20
echo "-- pip: index reachable? (5 s timeout, no retries)"
timeout 60 python3 -m pip install --disable-pip-version-check --timeout 5 --retries 0 --target /tmp/cv opencv-python-headless 2>&1 | tail -3
echo "-- pip: local wheel caches / find-links"
python3 -m pip config list 2>&1 | head; python3 -m pip cache list 2>&1 | grep -i -c opencv
ls /root/.cache/pip /wheelhouse /opt/wheelhouse /tmp/wheelhouse 2>&1 | head -5
echo "-- conda"
which conda mamba micromamba 2>&1 | head -3
timeout 60 conda install -y -p /tmp/cvenv --offline opencv 2>&1 | tail -2
echo "-- apt"
timeout 30 apt-get install -y --no-download python3-opencv 2>&1 | tail -2
echo "-- network at all?"
timeout 8 python3 -c 'import socket; socket.create_connection(("pypi.org", 443), 5); print("pypi.org:443 reachable")' 2>&1 | tail -1
echo "-- verdict"
if PYTHONPATH=/tmp/cv:$PYTHONPATH python3 -c 'import cv2' 2>/dev/null; then
  echo "cv2 IMPORTS: generating the fixture and running the cross-check"
  PYTHONPATH=/tmp/cv:$PYTHONPATH python3 tests/golden/gen_cv2_golden.py && cp tests/golden/cv2_match_template.json gpurun_out/
  PYTHONPATH=/tmp/cv:$PYTHONPATH python3 -m pytest tests/test_cv2_crosscheck.py tests/test_cv2_golden.py -q -s 2>&1 | tail -15
else
  echo "no route to cv2 on this machine: parity stays unpinned at the cv2 boundary"
fi
} > $out 2>&1
tail -5 $out
