// oracle/ref_main.cpp — TEST INFRASTRUCTURE ONLY.  Entry point of the reference CLI binary
// (oracle/_ref/templering_sfm_ref): forwards to the reference's own main() (T:1518), which
// ref_harness.cpp compiles under the name ref_main.
extern "C" int ref_main(int argc, char** argv);
int main(int argc, char** argv) { return ref_main(argc, argv); }
