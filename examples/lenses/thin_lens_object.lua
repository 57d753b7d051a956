-- Parameters kept in an object made while the script loads; the per-pixel callback calls its methods.
local Lens = {}
Lens.__index = Lens
function Lens.new(k1, k2, limit) return setmetatable({k1 = k1, k2 = k2, limit = limit}, Lens) end
function Lens:bend(r) return r * (1 + self.k1 * r * r + self.k2 * r * r * r * r) end      -- radial distortion polynomial
function Lens:visible(r) return r <= self.limit end

local lens = Lens.new(0.18, 0.05, 1.4)

max_fov = 200
max_vfov = 200
lens_width = 2 * lens.limit
lens_height = 2 * lens.limit
onload = "f_contain"

function lens_inverse(x, y)
   local p = {x = x, y = y, r = sqrt(x * x + y * y)}
   if not lens:visible(p.r) then return nil end
   if p.r == 0 then return 0, 0, 1 end
   local theta = lens:bend(p.r)
   if theta > pi then return nil end
   local s = sin(theta) / p.r
   return p.x * s, p.y * s, cos(theta)
end
