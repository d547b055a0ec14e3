"""Source compatibility of roskfpos_amd/csrc/kfpos_adaptor.h with the reference node.

The two statements of Posgenerator.cpp that touch Vector3 beyond plain member access are compiled VERBATIM against the
adaptor header with the reference's language level (-std=c++11, CMakeLists.txt:7-9):
  Posgenerator.cpp:397-399   msg.pose.covariance[i] = report.covarianceMatrix(i);   (Armadillo linear, column-major)
  Posgenerator.cpp:542       Vector3 pose = {NAN, NAN, NAN};                         (aggregate initialisation)
CPU only: the snippet is linked against libkfpos_hip.so but makes no call that needs a GPU.
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "roskfpos_amd", "csrc")

SNIPPET = r"""
#include <cmath>
#include <cstdio>
#include "kfpos_adaptor.h"
#include "kfpos_publish.h"
using namespace kfpos_host;

struct Msg { struct { double covariance[36]; } pose; };

/* Posgenerator.cpp:385-399, the covariance copy as written there */
static void publishPositionReport(Vector3 report, Msg &msg) {
    if (!std::isnan(report.x)) {
    for (int i = 0; i < 36; i++) {
      msg.pose.covariance[i] = report.covarianceMatrix(i);
    }
    }
}

int main() {
  Vector3 pose = {NAN, NAN, NAN};                       /* Posgenerator.cpp:542, verbatim */
  if (!std::isnan(pose.x) || !std::isnan(pose.z)) return 1;
  if (pose.rotW != 0.0 || pose.angularSpeedZ != 0.0) return 2;   /* members not named: value-initialised */
  if (pose.covarianceMatrix.n_elem != 0) return 3;

  /* what stateToPose of the 9-state filter builds (KalmanFilterTOAIMU.cpp:217-239) */
  Vector3 report = Vector3();
  report.covarianceMatrix.eye(9, 9, 0.01);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) report.covarianceMatrix(i, j) = 10 * i + j + 1;
  report.covarianceMatrix(0, 7) = 77.0;
  Msg msg;
  publishPositionReport(report, msg);
  /* the first 36 LINEAR elements of a 9x9 column-major matrix: columns 0..3 */
  for (int i = 0; i < 36; ++i) {
    const int r = i % 9, c = i / 9;
    const double want = (r < 3 && c < 3) ? 10 * r + c + 1 : (r == c ? 0.01 : 0.0);
    if (msg.pose.covariance[i] != want) { std::printf("linear %d: %g != %g\n", i, msg.pose.covariance[i], want); return 4; }
  }
  /* 6x6 (KalmanFilterTOA.cpp:159-183): the whole matrix, column-major */
  Vector3 six = Vector3();
  six.covarianceMatrix.zeros(6, 6);
  six.covarianceMatrix(1, 2) = 12.0;
  publishPositionReport(six, msg);
  if (msg.pose.covariance[2 * 6 + 1] != 12.0 || msg.pose.covariance[1 * 6 + 2] != 0.0) return 5;
  /* Armadillo's bounds check: an empty matrix throws std::logic_error on (i) */
  bool thrown = false;
  try { publishPositionReport(Vector3(), msg); } catch (const std::logic_error &) { thrown = true; }
  Vector3 empty = Vector3();
  empty.x = 1.0;
  try { publishPositionReport(empty, msg); } catch (const std::logic_error &) { thrown = true; }
  if (!thrown) return 6;
  /* the publisher mirror uses the same statement */
  PosePublisher pub;
  report.x = report.y = report.z = 1.0;
  if (!pub.publish(report, 0.0) || pub.msg.covariance[3 * 9 + 3] != 0.01) return 7;
  /* Beacon is the reference's aggregate too (sensor_types.h:19-23) */
  Beacon b = {7, 0, {1.0, 2.0, 3.0}};
  if (b.position.y != 2.0 || b.position.covarianceMatrix.n_rows != 0) return 8;
  std::printf("ok\n");
  return 0;
}
"""


def test_reference_statements_compile_with_cxx11(tmp_path):
    src = tmp_path / "compat.cpp"
    src.write_text(SNIPPET)
    exe = tmp_path / "compat"
    lib = os.path.join(CSRC, "libkfpos_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                           "-I", CSRC, "-o", str(exe), str(src), "-L", CSRC, "-lkfpos_hip",
                           "-Wl,-rpath," + CSRC])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.strip() == "ok"


def test_host_headers_compile_as_cxx11():
    """kfpos_replay.cpp (adaptor + ingest + publisher) at the reference's language level, syntax only."""
    subprocess.check_call(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(CSRC, "kfpos_replay.cpp")])
