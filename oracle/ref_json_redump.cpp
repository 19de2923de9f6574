// ref_json_redump.cpp -- a JSON file read and written back by the REAL nlohmann/json (3.1.1, /opt/conda/include/json.hpp in this image), exactly as
// Testbed::save_edits writes (`f << output_json << std::endl`, testbed.cu:3190-3204) and Testbed::load_edits reads (`i >> j`, :3209-3211).
// TEST INFRASTRUCTURE ONLY (oracle/Makefile -> oracle/_ref/json_redump where that header exists).  Why: oracle/_ref/libref_json.so compiles the reference's to_json
// code against a STAND-IN for <json/json.hpp> (oracle/ref_stubs/json/json.hpp: the submodule that vendors nlohmann/json is empty, and this image's 3.1.1 cannot take
// its place there -- json_binding.h calls basic_json::contains (3.6.0) and the snapshot writer json::binary (3.8.0)).  The stand-in prints floats with 17
// significant digits where nlohmann prints the shortest round-trip form.  This program turns the stand-in's text into the text the real library emits for the same
// value tree (its own number formatting, key order, `null`s): tests/test_ref_pin.py holds nrs_edits_open to it as well.
#include <json.hpp>

#include <fstream>
#include <iostream>

int main(int argc, char** argv) {
	if (argc != 3) { std::cerr << "usage: json_redump in.json out.json\n"; return 2; }
	std::ifstream i(argv[1]);
	if (!i) { std::cerr << "cannot open " << argv[1] << "\n"; return 1; }
	nlohmann::json j;
	i >> j;
	std::ofstream f(argv[2]);
	f << j << std::endl;
	return f ? 0 : 1;
}
