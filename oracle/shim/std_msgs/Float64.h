#pragma once
namespace std_msgs { struct Float64 { double data; }; }
