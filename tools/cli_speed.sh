python3 - <<'PY'
import numpy as np, struct
n = 44100 * 600
rng = np.random.RandomState(1)
s16 = rng.randint(-8000, 8000, size=(n, 2)).astype("<i2")
body = s16.tobytes()
open("/tmp/long.wav", "wb").write(b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack("<I", 16) + struct.pack("<HHIIHH", 1, 2, 44100, 176400, 4, 16) + b"data" + struct.pack("<I", len(body)) + body)
PY
for b in 256 2048 8192; do python3 -c "
import subprocess, time, sys
t = time.time()
subprocess.check_call(['./atracdenc_amd/at3hipenc', '-e', 'atrac3', '-i', '/tmp/long.wav', '-o', '/tmp/out_$b.oma', '--nostdout', '--batch', '$b'])
dt = time.time() - t
print('batch $b: %.2f s wall for 600 s of audio = %.0f x realtime (incl. process start, WAV read, PCIe, file write)' % (dt, 600 / dt))
"; done
cmp /tmp/out_256.oma /tmp/out_8192.oma && echo identical; ls -la /tmp/out_2048.oma
