// Stand-in for the reference's (empty) tinylogger submodule: the render-path sources only name tlog in code we do not build.
#pragma once
#include <iostream>
#include <string>
#include <vector>
